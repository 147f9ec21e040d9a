// Data-movement kernels of the ViT-ED path: casts, weight-shadow transposes, patch extraction,
// cls/pos rows and deterministic row sums.  All HBM-bound; loads/stores are coalesced and (where the
// shape allows) 16 B per lane.
#include "common.h"

// ------------------------------------------------------------------------------------------------
// cast
// ------------------------------------------------------------------------------------------------
template <typename S, typename D>
__global__ void cast_kernel(const S* __restrict__ src, D* __restrict__ dst, int64_t n) {
    int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
    for (; i + 3 < n; i += stride) {
        S a = src[i], b = src[i + 1], c = src[i + 2], d = src[i + 3];
        dst[i] = from_f32<D>(to_f32(a));
        dst[i + 1] = from_f32<D>(to_f32(b));
        dst[i + 2] = from_f32<D>(to_f32(c));
        dst[i + 3] = from_f32<D>(to_f32(d));
    }
    for (; i < n; ++i) dst[i] = from_f32<D>(to_f32(src[i]));  // only the thread that owns the ragged tail gets here
}

extern "C" int vited_cast(const void* src, int src_dtype, void* dst, int dst_dtype, int64_t n, void* stream) {
    if (!src || !dst || n < 0) return VITED_ERR_BAD_ARG;
    if (n == 0) return VITED_OK;
    hipStream_t s = (hipStream_t)stream;
    const int threads = 256;
    int64_t blocks = ceil_div64(n, threads * 4);
    if (blocks > 4096) blocks = 4096;
    if (src_dtype == VITED_F32 && dst_dtype == VITED_BF16)
        hipLaunchKernelGGL((cast_kernel<float, bf16>), dim3(blocks), dim3(threads), 0, s, (const float*)src, (bf16*)dst, n);
    else if (src_dtype == VITED_BF16 && dst_dtype == VITED_F32)
        hipLaunchKernelGGL((cast_kernel<bf16, float>), dim3(blocks), dim3(threads), 0, s, (const bf16*)src, (float*)dst, n);
    else if (src_dtype == VITED_F32 && dst_dtype == VITED_F32)
        hipLaunchKernelGGL((cast_kernel<float, float>), dim3(blocks), dim3(threads), 0, s, (const float*)src, (float*)dst, n);
    else if (src_dtype == VITED_BF16 && dst_dtype == VITED_BF16)
        hipLaunchKernelGGL((cast_kernel<bf16, bf16>), dim3(blocks), dim3(threads), 0, s, (const bf16*)src, (bf16*)dst, n);
    else
        return VITED_ERR_UNSUPPORTED;
    return vited_check_launch();
}

// ------------------------------------------------------------------------------------------------
// cast + transpose through a padded 32x32 LDS tile (both sides coalesced)
// ------------------------------------------------------------------------------------------------
template <typename D>
__global__ void cast_transpose_kernel(const float* __restrict__ src, D* __restrict__ dst, int64_t rows, int64_t cols) {
    __shared__ float tile[32][33];
    const int64_t c0 = (int64_t)blockIdx.x * 32, r0 = (int64_t)blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: 32 x 8
    for (int j = ty; j < 32; j += 8) {
        int64_t r = r0 + j, c = c0 + tx;
        tile[j][tx] = (r < rows && c < cols) ? src[r * cols + c] : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        int64_t c = c0 + j, r = r0 + tx;
        if (c < cols && r < rows) dst[c * rows + r] = from_f32<D>(tile[tx][j]);
    }
}

extern "C" int vited_cast_transpose(const float* src, void* dst, int dst_dtype, int64_t rows, int64_t cols, void* stream) {
    if (!src || !dst || rows <= 0 || cols <= 0) return VITED_ERR_BAD_ARG;
    dim3 grid((unsigned)ceil_div64(cols, 32), (unsigned)ceil_div64(rows, 32));
    hipStream_t s = (hipStream_t)stream;
    if (dst_dtype == VITED_BF16)
        hipLaunchKernelGGL((cast_transpose_kernel<bf16>), grid, dim3(256), 0, s, src, (bf16*)dst, rows, cols);
    else if (dst_dtype == VITED_F32)
        hipLaunchKernelGGL((cast_transpose_kernel<float>), grid, dim3(256), 0, s, src, (float*)dst, rows, cols);
    else
        return VITED_ERR_UNSUPPORTED;
    return vited_check_launch();
}

// ------------------------------------------------------------------------------------------------
// Multi-tensor weight-shadow refresh: every fp32 master weight [rows, cols] -> its bf16 [rows, cols]
// and / or transposed bf16 [cols, rows] shadow in ONE launch (was ~180 launches per optimizer step).
// desc[t] = {src, dst (0 = none), dst_t (0 = none), rows, cols, first_tile}; one workgroup per 64x64 tile.
// ------------------------------------------------------------------------------------------------
#define WS_DESC_WORDS 6
__global__ void __launch_bounds__(256)
cast_weights_kernel(const int64_t* __restrict__ desc, int count) {
    __shared__ bf16 tile[64][66];   // 33-dword rows: the transposed read walks 64 different banks
    const int64_t blk = blockIdx.x;
    int lo = 0, hi = count - 1;     // last descriptor whose first_tile <= blk
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (desc[(int64_t)mid * WS_DESC_WORDS + 5] <= blk) lo = mid; else hi = mid - 1;
    }
    const int64_t* d = desc + (int64_t)lo * WS_DESC_WORDS;
    const float* __restrict__ src = (const float*)d[0];
    bf16* __restrict__ dst = (bf16*)d[1];
    bf16* __restrict__ dst_t = (bf16*)d[2];
    const int64_t rows = d[3], cols = d[4];
    const int64_t t = blk - d[5], tiles_c = (cols + 63) >> 6;
    const int64_t r0 = (t / tiles_c) << 6, c0 = (t % tiles_c) << 6;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;   // 16 x 16: 4 columns x 4 rows per thread
    const bool vec = (cols & 3) == 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int lr = ty + 16 * j;
        const int64_t r = r0 + lr, c = c0 + tx * 4;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (r < rows) {
            if (vec && c + 3 < cols) {
                const f32x4 q = *(const f32x4*)(src + r * cols + c);
                v[0] = q[0]; v[1] = q[1]; v[2] = q[2]; v[3] = q[3];
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (c + e < cols) v[e] = src[r * cols + c + e];
            }
        }
        bf16 b[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) { b[e] = (bf16)v[e]; tile[lr][tx * 4 + e] = b[e]; }
        if (dst && r < rows) {
            if (vec && c + 3 < cols) *(bf16x4*)(dst + r * cols + c) = bf16x4{b[0], b[1], b[2], b[3]};
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (c + e < cols) dst[r * cols + c + e] = b[e];
            }
        }
    }
    if (!dst_t) return;
    __syncthreads();
    const bool vec_t = (rows & 3) == 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int lc = ty + 16 * j;                 // column of the source tile = row of the transposed shadow
        const int64_t c = c0 + lc, r = r0 + tx * 4;
        if (c >= cols) continue;
        bf16 b[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) b[e] = tile[tx * 4 + e][lc];
        if (vec_t && r + 3 < rows) *(bf16x4*)(dst_t + c * rows + r) = bf16x4{b[0], b[1], b[2], b[3]};
        else {
#pragma unroll
            for (int e = 0; e < 4; ++e) if (r + e < rows) dst_t[c * rows + r + e] = b[e];
        }
    }
}

extern "C" int vited_cast_weights(const int64_t* desc, int count, int64_t total_tiles, void* stream) {
    if (!desc || count <= 0 || total_tiles <= 0 || total_tiles > 0x7fffffff) return VITED_ERR_BAD_ARG;
    hipLaunchKernelGGL(cast_weights_kernel, dim3((unsigned)total_tiles), dim3(256), 0, (hipStream_t)stream, desc, count);
    return vited_check_launch();
}

// ------------------------------------------------------------------------------------------------
// patchify: one workgroup per (image, patch-row py): it reads p full image rows per channel
// (coalesced along x) and writes the G patch rows of the token matrix.
// ------------------------------------------------------------------------------------------------
template <typename D>
__global__ void patchify_kernel(const float* __restrict__ img, int64_t img_bs, const int64_t* __restrict__ bidx,
                                D* __restrict__ out, int chans, int S, int p) {
    const int G = S / p;
    const int64_t b = blockIdx.y;
    const int py = blockIdx.x;
    const int64_t src_b = bidx ? bidx[b] : b;
    const float* base = img + src_b * img_bs;
    const int Kp = chans * p * p;
    const int total = chans * p * S;  // elements in this strip: (c, i, x)
    D* obase = out + (b * G * G + (int64_t)py * G) * Kp;
    for (int e = threadIdx.x; e < total; e += blockDim.x) {
        const int x = e % S;
        const int ci = e / S;  // c * p + i
        const int i = ci % p, c = ci / p;
        const float v = base[((int64_t)c * S + (py * p + i)) * S + x];
        const int px = x / p, j = x - px * p;
        obase[(int64_t)px * Kp + (c * p + i) * p + j] = from_f32<D>(v);
    }
}

extern "C" int vited_patchify(const float* img, int64_t img_bs, const int64_t* batch_index, void* out, int out_dtype,
                              int64_t batch, int chans, int img_size, int patch, void* stream) {
    if (!img || !out || batch <= 0 || chans <= 0 || img_size <= 0 || patch <= 0 || img_size % patch) return VITED_ERR_BAD_ARG;
    dim3 grid(img_size / patch, (unsigned)batch);
    hipStream_t s = (hipStream_t)stream;
    if (out_dtype == VITED_BF16)
        hipLaunchKernelGGL((patchify_kernel<bf16>), grid, dim3(256), 0, s, img, img_bs, batch_index, (bf16*)out, chans, img_size, patch);
    else if (out_dtype == VITED_F32)
        hipLaunchKernelGGL((patchify_kernel<float>), grid, dim3(256), 0, s, img, img_bs, batch_index, (float*)out, chans, img_size, patch);
    else
        return VITED_ERR_UNSUPPORTED;
    return vited_check_launch();
}

// ------------------------------------------------------------------------------------------------
// patchify straight from uint8 pixels: ToTensor + Normalize(mean, std) of the input pipeline (data/transforms.py:14-18) folded
// into the patch extraction, so a batch crosses PCIe and HBM as 1 byte per pixel instead of 4.
// value = pixel * scale[c] + shift[c] with scale = 1 / (255 std), shift = -mean / std.
// ------------------------------------------------------------------------------------------------
struct U8Norm { float scale[4], shift[4]; };

template <typename D>
__global__ void patchify_u8_kernel(const uint8_t* __restrict__ img, int64_t img_bs, const int64_t* __restrict__ bidx,
                                   D* __restrict__ out, int chans, int S, int p, U8Norm nrm) {
    const int G = S / p;
    const int64_t b = blockIdx.y;
    const int py = blockIdx.x;
    const int64_t src_b = bidx ? bidx[b] : b;
    const uint8_t* base = img + src_b * img_bs;
    const int Kp = chans * p * p;
    const int total = chans * p * S;  // elements in this strip: (c, i, x)
    D* obase = out + (b * G * G + (int64_t)py * G) * Kp;
    for (int e = threadIdx.x; e < total; e += blockDim.x) {
        const int x = e % S;
        const int ci = e / S;  // c * p + i
        const int i = ci % p, c = ci / p;
        const float v = fmaf((float)base[((int64_t)c * S + (py * p + i)) * S + x], nrm.scale[c], nrm.shift[c]);
        const int px = x / p, j = x - px * p;
        obase[(int64_t)px * Kp + (c * p + i) * p + j] = from_f32<D>(v);
    }
}

extern "C" int vited_patchify_u8(const uint8_t* img, int64_t img_bs, const int64_t* batch_index, void* out, int out_dtype,
                                 int64_t batch, int chans, int img_size, int patch, const float* mean, const float* std,
                                 void* stream) {
    if (!img || !out || !mean || !std || batch <= 0 || chans <= 0 || chans > 4 || img_size <= 0 || patch <= 0 || img_size % patch)
        return VITED_ERR_BAD_ARG;
    U8Norm nrm = {};
    for (int c = 0; c < chans; ++c) {
        if (!(std[c] > 0.f)) return VITED_ERR_BAD_ARG;
        nrm.scale[c] = 1.0f / (255.0f * std[c]);
        nrm.shift[c] = -mean[c] / std[c];
    }
    dim3 grid(img_size / patch, (unsigned)batch);
    hipStream_t s = (hipStream_t)stream;
    if (out_dtype == VITED_BF16)
        hipLaunchKernelGGL((patchify_u8_kernel<bf16>), grid, dim3(256), 0, s, img, img_bs, batch_index, (bf16*)out, chans, img_size, patch, nrm);
    else if (out_dtype == VITED_F32)
        hipLaunchKernelGGL((patchify_u8_kernel<float>), grid, dim3(256), 0, s, img, img_bs, batch_index, (float*)out, chans, img_size, patch, nrm);
    else
        return VITED_ERR_UNSUPPORTED;
    return vited_check_launch();
}

// ------------------------------------------------------------------------------------------------
// patch-pair assembly on the device (data/datasets/div2k_patch.py:108-121,155-162)
//   the host hands over ONE uint8 region [C, 2 S, 3 S] per sample (the RandomCrop / CenterCrop of the augmented image) and the
//   two grid cells its pair consists of; here every cell is eroded (CenterCrop(e)) and resized back to S x S exactly as
//   torchvision's Resize does it on a PIL image: Pillow's 8-bit bilinear resample = horizontal pass, then vertical pass, each
//   with two taps, 22-bit fixed-point coefficients and a uint8 intermediate (libImaging/Resample.c).  Bit-exact with Pillow
//   (oracle/pair_crops.py pins the restatement against Pillow itself).
// ------------------------------------------------------------------------------------------------
struct ResampleTaps {
    int x0, n;      // first tap, number of taps (1 or 2)
    int k0, k1;     // fixed-point weights (sum = 2^22 up to rounding)
};

__device__ __forceinline__ ResampleTaps resample_taps(int xx, int in_size, int out_size) {
    constexpr int PRECISION_BITS = 32 - 8 - 2;
    const double scale = (double)in_size / (double)out_size;    // <= 1 (erosion shrinks the cell): filter support 1
    const double center = (xx + 0.5) * scale;
    int xmin = (int)(center - 1.0 + 0.5);
    xmin = xmin < 0 ? 0 : xmin;
    int xmax = (int)(center + 1.0 + 0.5);
    xmax = (xmax < in_size ? xmax : in_size) - xmin;
    double w0 = 1.0 - fabs((double)xmin - center + 0.5), w1 = xmax > 1 ? 1.0 - fabs((double)xmin + 1.0 - center + 0.5) : 0.0;
    w0 = w0 < 0.0 ? 0.0 : w0;
    w1 = w1 < 0.0 ? 0.0 : w1;
    const double ww = w0 + w1;
    ResampleTaps t;
    t.x0 = xmin;
    t.n = xmax > 1 ? 2 : 1;
    t.k0 = (int)(0.5 + (w0 / ww) * (double)(1 << PRECISION_BITS));
    t.k1 = (int)(0.5 + (w1 / ww) * (double)(1 << PRECISION_BITS));
    return t;
}

__device__ __forceinline__ int resample_px(int p0, int p1, const ResampleTaps& t) {
    constexpr int PRECISION_BITS = 32 - 8 - 2;
    int acc = (1 << (PRECISION_BITS - 1)) + p0 * t.k0 + (t.n > 1 ? p1 * t.k1 : 0);
    acc >>= PRECISION_BITS;
    return acc < 0 ? 0 : (acc > 255 ? 255 : acc);
}

__global__ void __launch_bounds__(256)
crop_pairs_u8_kernel(const uint8_t* __restrict__ src, int64_t src_bs, const int* __restrict__ cells, const int* __restrict__ erode,
                     uint8_t* __restrict__ out, int chans, int S) {
    const int64_t b = blockIdx.z;
    const int img = blockIdx.y / chans, c = blockIdx.y % chans;
    int cell = cells[b * 2 + img], e = erode[b];
    cell = cell < 0 ? 0 : (cell > 5 ? 5 : cell);             // device-side arguments: clamp instead of reading out of bounds
    e = e < 1 ? 1 : (e > S ? S : e);
    const int off = (int)rint((S - e) / 2.0);                 // torchvision center_crop: int(round(.)), round half to even
    const int r0 = (cell / 3) * S + off, c0 = (cell % 3) * S + off;
    const uint8_t* base = src + b * src_bs + ((int64_t)c * 2 * S + r0) * (3 * S) + c0;   // eroded cell, row stride 3 S
    uint8_t* o = out + (((b * 2 + img) * chans + c) * (int64_t)S) * S;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < S * S; i += gridDim.x * blockDim.x) {
        const int y = i / S, x = i - y * S;
        const ResampleTaps tx = resample_taps(x, e, S), ty = resample_taps(y, e, S);
        const uint8_t* ra = base + (int64_t)ty.x0 * (3 * S) + tx.x0;
        const int h0 = resample_px(ra[0], tx.n > 1 ? ra[1] : 0, tx);           // horizontal pass on the two source rows
        int h1 = 0;
        if (ty.n > 1) {
            const uint8_t* rb = ra + 3 * S;
            h1 = resample_px(rb[0], tx.n > 1 ? rb[1] : 0, tx);
        }
        o[i] = (uint8_t)resample_px(h0, h1, ty);                                // vertical pass on the uint8 intermediate
    }
}

extern "C" int vited_crop_pairs_u8(const uint8_t* src, int64_t src_bs, const int* cells, const int* erode, uint8_t* out, int64_t batch,
                                   int chans, int img_size, void* stream) {
    if (!src || !cells || !erode || !out || batch <= 0 || chans <= 0 || img_size <= 0 || batch > 65535) return VITED_ERR_BAD_ARG;
    if (src_bs < (int64_t)chans * 6 * img_size * img_size) return VITED_ERR_BAD_ARG;
    const int per = img_size * img_size;
    dim3 grid((unsigned)((per + 1023) / 1024 < 8 ? (per + 1023) / 1024 : 8), (unsigned)(2 * chans), (unsigned)batch);
    hipLaunchKernelGGL(crop_pairs_u8_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, src_bs, cells, erode, out, chans, img_size);
    return vited_check_launch();
}

// ------------------------------------------------------------------------------------------------
// slice rows + cast
// ------------------------------------------------------------------------------------------------
template <typename D>
__global__ void slice_rows_cast_kernel(const float* __restrict__ in, D* __restrict__ out, int64_t in_rows,
                                       int64_t row_offset, int64_t rows, int64_t dim, int64_t total) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t per_b = rows * dim;
    for (; i < total; i += stride) {
        const int64_t b = i / per_b, rem = i - b * per_b;
        out[i] = from_f32<D>(in[(b * in_rows + row_offset) * dim + rem]);
    }
}

extern "C" int vited_slice_rows_cast(const float* in, void* out, int out_dtype, int64_t batch, int64_t in_rows,
                                     int64_t row_offset, int64_t rows, int64_t dim, void* stream) {
    if (!in || !out || batch <= 0 || rows <= 0 || dim <= 0 || row_offset < 0 || row_offset + rows > in_rows) return VITED_ERR_BAD_ARG;
    const int64_t total = batch * rows * dim;
    int64_t blocks = ceil_div64(total, 256);
    if (blocks > 8192) blocks = 8192;
    hipStream_t s = (hipStream_t)stream;
    if (out_dtype == VITED_BF16)
        hipLaunchKernelGGL((slice_rows_cast_kernel<bf16>), dim3(blocks), dim3(256), 0, s, in, (bf16*)out, in_rows, row_offset, rows, dim, total);
    else if (out_dtype == VITED_F32)
        hipLaunchKernelGGL((slice_rows_cast_kernel<float>), dim3(blocks), dim3(256), 0, s, in, (float*)out, in_rows, row_offset, rows, dim, total);
    else
        return VITED_ERR_UNSUPPORTED;
    return vited_check_launch();
}

// ------------------------------------------------------------------------------------------------
// cls row
// ------------------------------------------------------------------------------------------------
__global__ void write_cls_row_kernel(float* __restrict__ x, const float* __restrict__ cls, const float* __restrict__ pos,
                                     int64_t batch, int64_t rows_per_batch, int64_t dim) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = batch * dim;
    if (i >= total) return;
    const int64_t b = i / dim, d = i - b * dim;
    x[b * rows_per_batch * dim + d] = cls[d] + pos[d];
}

extern "C" int vited_write_cls_row(float* x, const float* cls, const float* pos, int64_t batch, int64_t rows_per_batch,
                                   int64_t dim, void* stream) {
    if (!x || !cls || !pos || batch <= 0 || rows_per_batch <= 0 || dim <= 0) return VITED_ERR_BAD_ARG;
    const int64_t total = batch * dim;
    hipLaunchKernelGGL(write_cls_row_kernel, dim3((unsigned)ceil_div64(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       x, cls, pos, batch, rows_per_batch, dim);
    return vited_check_launch();
}

// ------------------------------------------------------------------------------------------------
// deterministic sum over the leading dimension: out[r] = sum_b in[b, r]
// pass 1: grid (ceil(width/256), S) -> partial[s, r]; pass 2: same kernel over the S partial rows.
// ------------------------------------------------------------------------------------------------
template <typename S_>
__global__ void sum_rows_kernel(const S_* __restrict__ in, int64_t in_ld, float* __restrict__ out, int64_t batch,
                                int64_t width, int64_t rows_per_split) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= width) return;
    const int64_t b0 = (int64_t)blockIdx.y * rows_per_split;
    int64_t b1 = b0 + rows_per_split;
    if (b1 > batch) b1 = batch;
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
    int64_t b = b0;
    for (; b + 3 < b1; b += 4) {
        acc0 += to_f32(in[b * in_ld + r]);
        acc1 += to_f32(in[(b + 1) * in_ld + r]);
        acc2 += to_f32(in[(b + 2) * in_ld + r]);
        acc3 += to_f32(in[(b + 3) * in_ld + r]);
    }
    for (; b < b1; ++b) acc0 += to_f32(in[b * in_ld + r]);
    out[(int64_t)blockIdx.y * width + r] = (acc0 + acc1) + (acc2 + acc3);
}

static inline int64_t sum_rows_splits(int64_t batch, int64_t width) {
    // aim for >= ~1024 workgroups, at least 16 rows per split
    const int64_t col_blocks = ceil_div64(width, 256);
    int64_t s = ceil_div64(1024, col_blocks);
    const int64_t max_s = ceil_div64(batch, 16);
    if (s > max_s) s = max_s;
    if (s < 1) s = 1;
    return s;
}

extern "C" int64_t vited_sum_rows_workspace_bytes(int64_t batch, int64_t width) {
    return sum_rows_splits(batch, width) * width * (int64_t)sizeof(float);
}

extern "C" int vited_sum_rows(const void* in, int in_dtype, int64_t in_ld, float* out, int64_t batch, int64_t width,
                              float* workspace, int64_t workspace_bytes, void* stream) {
    if (!in || !out || batch <= 0 || width <= 0 || in_ld < width) return VITED_ERR_BAD_ARG;
    const int64_t S = sum_rows_splits(batch, width);
    hipStream_t s = (hipStream_t)stream;
    const unsigned col_blocks = (unsigned)ceil_div64(width, 256);
    float* pass1_out = out;
    if (S > 1) {
        if (!workspace || workspace_bytes < S * width * (int64_t)sizeof(float)) return VITED_ERR_WORKSPACE;
        pass1_out = workspace;
    }
    const int64_t rps = ceil_div64(batch, S);
    if (in_dtype == VITED_F32)
        hipLaunchKernelGGL((sum_rows_kernel<float>), dim3(col_blocks, (unsigned)S), dim3(256), 0, s, (const float*)in, in_ld, pass1_out, batch, width, rps);
    else if (in_dtype == VITED_BF16)
        hipLaunchKernelGGL((sum_rows_kernel<bf16>), dim3(col_blocks, (unsigned)S), dim3(256), 0, s, (const bf16*)in, in_ld, pass1_out, batch, width, rps);
    else
        return VITED_ERR_UNSUPPORTED;
    if (S > 1)
        hipLaunchKernelGGL((sum_rows_kernel<float>), dim3(col_blocks, 1), dim3(256), 0, s, (const float*)workspace, width, out, S, width, S);
    return vited_check_launch();
}

// out[r] (+)= sum_b in[b, r] in one pass; accumulate = add onto what out already holds (gradient accumulation
// straight into a parameter's .grad, see vited_linear_bwd_weight)
__device__ __forceinline__ void sum_slabs_body(const float* __restrict__ in, int64_t in_ld, float* __restrict__ out, int64_t batch,
                                               int64_t width, int accumulate, int64_t r) {
    if (r >= width) return;
    if (r + 3 < width) {
        f32x4 acc = accumulate ? *(const f32x4*)(out + r) : f32x4{0.f, 0.f, 0.f, 0.f};
        int64_t b = 0;
        for (; b + 4 <= batch; b += 4) {   // four independent 16-B loads in flight per lane
            const f32x4 v0 = *(const f32x4*)(in + b * in_ld + r), v1 = *(const f32x4*)(in + (b + 1) * in_ld + r);
            const f32x4 v2 = *(const f32x4*)(in + (b + 2) * in_ld + r), v3 = *(const f32x4*)(in + (b + 3) * in_ld + r);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] += (v0[e] + v1[e]) + (v2[e] + v3[e]);
        }
        for (; b < batch; ++b) {
            const f32x4 v = *(const f32x4*)(in + b * in_ld + r);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] += v[e];
        }
        *(f32x4*)(out + r) = acc;
    } else {
        for (int64_t c = r; c < width; ++c) {
            float acc = accumulate ? out[c] : 0.f;
            for (int64_t b = 0; b < batch; ++b) acc += in[b * in_ld + c];
            out[c] = acc;
        }
    }
}

__global__ void sum_slabs_kernel(const float* __restrict__ in, int64_t in_ld, float* __restrict__ out, int64_t batch, int64_t width,
                                 int accumulate) {
    sum_slabs_body(in, in_ld, out, batch, width, accumulate, ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4);
}

// two slab sets of one split-K launch (dW and dbias) reduced by ONE launch: blocks [0, blocks_a) take set A
__global__ void sum_slabs_pair_kernel(const float* __restrict__ in_a, int64_t ld_a, float* __restrict__ out_a, int64_t width_a,
                                      int blocks_a, const float* __restrict__ in_b, int64_t ld_b, float* __restrict__ out_b,
                                      int64_t width_b, int64_t batch, int accumulate) {
    if ((int)blockIdx.x < blocks_a)
        sum_slabs_body(in_a, ld_a, out_a, batch, width_a, accumulate, ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4);
    else
        sum_slabs_body(in_b, ld_b, out_b, batch, width_b, accumulate,
                       ((int64_t)(blockIdx.x - blocks_a) * blockDim.x + threadIdx.x) * 4);
}


// any width / alignment (e.g. the [1, 384] head of config H and its 1-element bias): one column per thread
__global__ void sum_slabs_scalar_kernel(const float* __restrict__ in, int64_t in_ld, float* __restrict__ out, int64_t batch,
                                        int64_t width, int accumulate) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= width) return;
    float acc = accumulate ? out[c] : 0.f;
    for (int64_t b = 0; b < batch; ++b) acc += in[b * in_ld + c];
    out[c] = acc;
}

int sum_rows_f32_single_pass(const float* in, int64_t in_ld, float* out, int64_t batch, int64_t width, hipStream_t s, int accumulate) {
    const bool vec = (width % 4 == 0) && (in_ld % 4 == 0) && ((((uintptr_t)in) | ((uintptr_t)out)) & 15) == 0;
    if (vec)
        hipLaunchKernelGGL(sum_slabs_kernel, dim3((unsigned)ceil_div64(width, 1024)), dim3(256), 0, s, in, in_ld, out, batch, width, accumulate);
    else
        hipLaunchKernelGGL(sum_slabs_scalar_kernel, dim3((unsigned)ceil_div64(width, 256)), dim3(256), 0, s, in, in_ld, out, batch, width, accumulate);
    return vited_check_launch();
}

// The slabs of a batched weight-gradient launch (gemm_tn_batch): one split's block = the products' [N, K] slabs back to back
// (slab_stride floats), bias sums likewise (bias_stride floats).  One launch sums every product's slabs onto ITS dW / dbias:
// a thread owns 4 consecutive floats of the concatenated width (every product's width is a multiple of 4).
#define SB_MAX 40
struct SumBatch {
    float* out[2 * SB_MAX];        // dW of product i, then dbias of product i (null = none)
    int64_t start[2 * SB_MAX + 1]; // first float of each section in the concatenated space [slabs | bias slabs]
    const float* slabs;
    const float* bias_slabs;
    int64_t slab_stride, bias_stride, splits;
    int sections, accumulate;
};

__global__ void __launch_bounds__(256) sum_slabs_batch_kernel(const SumBatch b) {
    const int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (r >= b.start[b.sections]) return;
    int sec = 0;
    for (int i = 1; i < b.sections; ++i)
        if (b.start[i] <= r) sec = i;
    float* out = b.out[sec];
    if (!out) return;
    const bool is_bias = r >= b.slab_stride;
    const float* in = is_bias ? b.bias_slabs + (r - b.slab_stride) : b.slabs + r;
    const int64_t ld = is_bias ? b.bias_stride : b.slab_stride;
    out += r - b.start[sec];
    f32x4 acc = b.accumulate ? *(const f32x4*)out : f32x4{0.f, 0.f, 0.f, 0.f};
    int64_t k = 0;
    for (; k + 4 <= b.splits; k += 4) {
        const f32x4 v0 = *(const f32x4*)(in + k * ld), v1 = *(const f32x4*)(in + (k + 1) * ld);
        const f32x4 v2 = *(const f32x4*)(in + (k + 2) * ld), v3 = *(const f32x4*)(in + (k + 3) * ld);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] += (v0[e] + v1[e]) + (v2[e] + v3[e]);
    }
    for (; k < b.splits; ++k) {
        const f32x4 v = *(const f32x4*)(in + k * ld);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] += v[e];
    }
    *(f32x4*)out = acc;
}

int sum_slabs_batch(int count, const float* slabs, int64_t slab_stride, const float* bias_slabs, int64_t bias_stride, int64_t splits,
                    const int64_t* widths, float* const* dW, const int64_t* nbias, float* const* dbias, int accumulate, hipStream_t s) {
    if (count < 1 || count > SB_MAX) return VITED_ERR_BAD_ARG;
    SumBatch b = {};
    int64_t off = 0;
    for (int i = 0; i < count; ++i) {
        if (widths[i] % 4 || nbias[i] % 4 || ((uintptr_t)dW[i] & 15) || ((uintptr_t)dbias[i] & 15)) return VITED_ERR_UNSUPPORTED;
        b.out[i] = dW[i];
        b.start[i] = off;
        off += widths[i];
    }
    if (off != slab_stride) return VITED_ERR_BAD_ARG;
    for (int i = 0; i < count; ++i) {
        b.out[count + i] = dbias[i];
        b.start[count + i] = off;
        off += nbias[i];
    }
    b.start[2 * count] = off;
    b.sections = 2 * count;
    b.slabs = slabs; b.bias_slabs = bias_slabs; b.slab_stride = slab_stride; b.bias_stride = bias_stride; b.splits = splits;
    b.accumulate = accumulate;
    if (((uintptr_t)slabs | (uintptr_t)bias_slabs) & 15 || slab_stride % 4 || bias_stride % 4) return VITED_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(sum_slabs_batch_kernel, dim3((unsigned)ceil_div64(off, 1024)), dim3(256), 0, s, b);
    return vited_check_launch();
}

int sum_slabs_pair(const float* in_a, int64_t ld_a, float* out_a, int64_t width_a, const float* in_b, int64_t ld_b, float* out_b,
                   int64_t width_b, int64_t batch, hipStream_t s, int accumulate) {
    const bool vec = (width_a % 4 == 0) && (ld_a % 4 == 0) && (width_b % 4 == 0) && (ld_b % 4 == 0) &&
                     ((((uintptr_t)in_a) | ((uintptr_t)out_a) | ((uintptr_t)in_b) | ((uintptr_t)out_b)) & 15) == 0;
    if (!vec) {
        int rc = sum_rows_f32_single_pass(in_a, ld_a, out_a, batch, width_a, s, accumulate);
        if (rc == VITED_OK) rc = sum_rows_f32_single_pass(in_b, ld_b, out_b, batch, width_b, s, accumulate);
        return rc;
    }
    const int ba = (int)ceil_div64(width_a, 1024), bb = (int)ceil_div64(width_b, 1024);
    hipLaunchKernelGGL(sum_slabs_pair_kernel, dim3((unsigned)(ba + bb)), dim3(256), 0, s, in_a, ld_a, out_a, width_a, ba, in_b, ld_b,
                       out_b, width_b, batch, accumulate);
    return vited_check_launch();
}
