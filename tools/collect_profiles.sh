#!/bin/bash
# copies the summaries of a tools/r6_final.sh run from gpurun_out/<tag>/ (scratch) into profiles/ (tracked)
# usage: tools/collect_profiles.sh <tag> <round prefix, e.g. r06>
set -u
T=gpurun_out/$1; R=$2
tail -1 $T/bench.json > profiles/${R}_bench_line.json
cp $T/bench_detail.json profiles/${R}_bench.json
cp $T/steady_vil_small_224.txt profiles/${R}_bench_small224_steady_state.txt
cp $T/steady_vil_medium_deep_384.txt profiles/${R}_bench_meddeep384_steady_state.txt
cp $T/kernel_stats_vil_small_224.csv profiles/${R}_bench_small224_kernel_stats.csv
cp $T/kernel_stats_vil_medium_deep_384.csv profiles/${R}_bench_meddeep384_kernel_stats.csv
cp $T/pmcstep/pmc_traffic.json profiles/${R}_pmc_traffic.json
cp $T/parity_report.txt profiles/${R}_parity_report.txt
cp $T/ab_summary.txt profiles/${R}_attn_ab_vs_round5.txt
cp $T/cw_vs_wave.txt profiles/${R}_cw_vs_wave.txt 2>/dev/null
cat $T/ab_bench_nocw.txt $T/ab_bench_round5.txt > profiles/${R}_step_ab.txt 2>/dev/null
cp $T/dense_bench.txt profiles/${R}_dense_bench.txt 2>/dev/null
cp $T/dma_rate.txt profiles/${R}_dma_rate_ubench.txt 2>/dev/null
cp $T/lds_atomic.txt profiles/${R}_lds_atomic_ubench.txt 2>/dev/null
cp $T/pmc/pipe_utilisation.txt profiles/${R}_pipe_utilisation.txt 2>/dev/null
rm -f profiles/${R}_pmc_k_*.json
python tools/pmc_all_summary.py $T/pmc profiles/${R}_pmc_ > /dev/null 2>&1
ls profiles | grep "^${R}_" | wc -l
