#!/bin/bash
# One round's measurement set on the GPU box (run through gpurun from the repo root):
#   scripts/profile_round.sh r01e
# writes gpurun_out/<tag>_*: the bench line, the same under rocprofv3 --kernel-trace --stats, and the two PMC passes
# (FETCH_SIZE, WRITE_SIZE; separate runs, kernel-trace only) reduced to per-launch HBM bytes.
set -e -o pipefail
tag=${1:-r01x}
root=$(pwd)
out=$root/gpurun_out
mkdir -p $out
python bench.py > $out/${tag}_bench_n1.json 2> $out/${tag}_bench_n1.err
echo "bench done"
cd /tmp && export TMPDIR=/tmp
rm -rf $out/${tag}_stats $out/${tag}_pmc_fetch $out/${tag}_pmc_write
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats -- python3 $root/bench.py --no-cpu-baseline --no-configs > $out/${tag}_bench_under_rocprof_n1.json 2> $out/${tag}_stats.err
echo "stats done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/${tag}_pmc_fetch -- python3 $root/bench.py --steps 60 --warmup 20 --min-steps 60 --no-cpu-baseline --no-configs > $out/${tag}_pmc_fetch.json 2> $out/${tag}_pmc_fetch.err
echo "fetch done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/${tag}_pmc_write -- python3 $root/bench.py --steps 60 --warmup 20 --min-steps 60 --no-cpu-baseline --no-configs > $out/${tag}_pmc_write.json 2> $out/${tag}_pmc_write.err
echo "write done"
cd $root
python scripts/pmc_traffic.py $out/${tag}_pmc_fetch $out/${tag}_pmc_write $out/${tag}_traffic.json "$tag: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) -- python3 bench.py --steps 60 --warmup 20 --min-steps 60 --no-cpu-baseline --no-configs"
python scripts/pmc_summary.py $out/${tag}_pmc_fetch $out/${tag}_pmc_write > $out/${tag}_pmc_summary.txt
cp $(ls $out/${tag}_stats/*/*kernel_stats.csv | head -1) $out/${tag}_kernel_stats.csv
# the raw counter CSVs are large: keep only the summaries
rm -rf $out/${tag}_pmc_fetch $out/${tag}_pmc_write $out/${tag}_stats
ls -la $out/${tag}_*
