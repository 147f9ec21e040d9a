#!/bin/bash
# Collects the round's profile evidence on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 1200 -- 'bash profiles/collect.sh r03'
# Writes raw output under gpurun_out/prof_<tag>/ ; profiles/summarize.py turns it into the committed CSVs.
# Separate passes: kernel trace + stats, then one --pmc pass per counter (FETCH_SIZE and WRITE_SIZE do not fit one pass).
set -e
TAG=${1:-r03}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$OLDPWD"
python3 bench.py > "$OUT/bench_n1.json" 2> "$OUT/bench_n1.err"
tail -1 "$OUT/bench_n1.json"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-workloads > "$OUT/stats.log" 2>&1
echo "stats pass done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-other-workloads --no-graph > "$OUT/pmc_fetch.log" 2>&1
echo "fetch pass done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-other-workloads --no-graph > "$OUT/pmc_write.log" 2>&1
echo "write pass done"
# keep only what summarize.py needs (the merged-back directory is capped at 64 MiB)
find "$OUT" -name '*_kernel_trace.csv' -path '*stats*' -size +30M -delete || true
ls -la "$OUT" "$OUT"/*/* | head -40
