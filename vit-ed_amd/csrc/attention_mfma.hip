// bf16 MFMA attention kernels (placeholder until the tiled kernels land: reports "unsupported" so the
// dispatcher uses the portable kernels).
#include "attention_kernels.h"

bool attention_mfma_supported(const AttnArgs&, bool) { return false; }
int attention_fwd_mfma(const AttnArgs&, hipStream_t) { return VITED_ERR_UNSUPPORTED; }
int attention_bwd_mfma(const AttnArgs&, hipStream_t) { return VITED_ERR_UNSUPPORTED; }
