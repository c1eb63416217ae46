#!/bin/bash
# Everything profiles/r02_* is made of, in one go on the GPU box:  gpurun -- 'bash tools/collect_profiles.sh'
# Output lands in gpurun_out/prof/ (copy what should be judged into profiles/).  Every step runs under its own timeout.
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
py() { timeout ${T:-300} python "$@"; }
T=400 py $R/bench.py --steps 480 --warmup 48 > $O/bench_480.json 2> $O/bench_480.err
py $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_20.json 2>/dev/null
py $R/bench.py --config extract --steps 480 --warmup 48 --no-cpu-baseline > $O/bench_extract.json 2>/dev/null
T=400 py $R/bench.py --config dense128 --steps 20 --warmup 3 > $O/bench_dense128.json 2>/dev/null
# kernel trace of the default bench
rm -rf /tmp/kb; T=400 timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/kb -o kb -- python $R/bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2>/dev/null
python $R/tools/prof_summary.py /tmp/kb/kb_results.db "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline" > $O/kernel_stats_bench.txt 2>&1
python $R/tools/timeline.py /tmp/kb/kb_results.db 1700 400 > $O/timeline_bench.txt 2>&1
# one batch of 8 frames per launch set, one stream: every kernel alone on the GPU
rm -rf /tmp/k1; CAELO_PIPE_STREAMS=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/k1 -o k1 -- python $R/tools/match_time.py > /dev/null 2>&1
python $R/tools/prof_summary.py /tmp/k1/k1_results.db "CAELO_PIPE_STREAMS=1 rocprofv3 --kernel-trace --stats -- python tools/match_time.py (8 frames per launch, one stream)" > $O/kernel_stats_one_stream.txt 2>&1
# the launches bench.py's roofline object times: 8 frames per launch (headline: the pipeline's launch shape) and one frame
rm -rf /tmp/rl; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/rl -o rl -- python $R/tools/roofline_launch.py 30 8 > /dev/null 2>&1
python $R/tools/prof_summary.py /tmp/rl/rl_results.db "rocprofv3 --kernel-trace --stats -- python tools/roofline_launch.py 30 8   (8 frames = 24576 patches per launch)" > $O/kernel_stats_roofline_launch.txt 2>&1
rm -rf /tmp/rl1; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/rl1 -o rl -- python $R/tools/roofline_launch.py 40 1 > /dev/null 2>&1
python $R/tools/prof_summary.py /tmp/rl1/rl_results.db "rocprofv3 --kernel-trace --stats -- python tools/roofline_launch.py 40 1   (one frame = 3072 patches per launch)" > $O/kernel_stats_roofline_launch_1frame.txt 2>&1
for c in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES; do
  rm -rf /tmp/pm_$c; timeout 300 rocprofv3 --kernel-trace --pmc $c -d /tmp/pm_$c -o pm -- python $R/tools/roofline_launch.py 10 8 > /dev/null 2>&1
done
python $R/tools/pmc_traffic_json.py /tmp/pm_FETCH_SIZE/pm_results.db /tmp/pm_WRITE_SIZE/pm_results.db "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python tools/roofline_launch.py 10 8; KB per launch of 8 frames (24576 patches), uncorrected" > $O/pmc_traffic.json 2>&1
( python $R/tools/pmc_summary.py /tmp/pm_SQ_VALU_MFMA_BUSY_CYCLES/pm_results.db k_enc; python $R/tools/pmc_summary.py /tmp/pm_SQ_BUSY_CYCLES/pm_results.db k_enc ) > $O/pmc_mfma_busy.txt 2>&1
# match kernel counters (one stream, 8 pairs per launch)
: > $O/pmc_match.txt
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY"; do
  rm -rf /tmp/pmm; CAELO_PIPE_STREAMS=1 timeout 300 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pmm -o pm -- python $R/tools/match_time.py > /dev/null 2>&1
  python $R/tools/pmc_summary.py /tmp/pmm/pm_results.db k_match >> $O/pmc_match.txt 2>&1
done
# stage 1, the two kernels' counters on the one-frame launch
: > $O/pmc_stage1_variants.txt
for w in 0 1; do
  echo "== CAELO_ENC_WAVE=$w" >> $O/pmc_stage1_variants.txt
  for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_BRANCH"; do
    rm -rf /tmp/pms; CAELO_ENC_WAVE=$w timeout 300 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pms -o pm -- python $R/tools/roofline_launch.py 8 > /dev/null 2>&1
    python $R/tools/pmc_summary.py /tmp/pms/pm_results.db k_enc_stage1 >> $O/pmc_stage1_variants.txt 2>&1
  done
  rm -rf /tmp/rls; CAELO_ENC_WAVE=$w timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/rls -o rl -- python $R/tools/roofline_launch.py 20 > /dev/null 2>&1
  python $R/tools/prof_summary.py /tmp/rls/rl_results.db 2>&1 | grep k_enc_stage1 | head -1 >> $O/pmc_stage1_variants.txt
done
# Dense(200): SQ counters of the pipelined kernel and of the round-1 kernel (CAELO_D1_PLAIN=1) on the 8-frame launch, and the
# HIP-event table of the four encoder kernels for 1 / 8 frames per launch
bash $R/tools/pmc_dense1.sh > /dev/null 2>&1; cp $R/gpurun_out/d1pmc.txt $O/pmc_dense1.txt
( py $R/tools/enc_table.py 2>&1 | tail -2; echo "CAELO_D1_PLAIN=1:"; CAELO_D1_PLAIN=1 py $R/tools/enc_table.py 2>&1 | tail -2 ) > $O/enc_table.txt
# the pair stage under three streams: the shipped library, and one built WITH packed-f32 instructions (if present)
( cd $R && timeout 300 python tools/stress_pairs.py 12 2>&1 | tail -1 ) > $O/stress_pairs.txt
if [ -f $R/tools/_variant_packed_f32.so ]; then
  ( cd $R && echo "PACKED_F32=1 build:" && CAELO_LIB=$R/tools/_variant_packed_f32.so timeout 300 python tools/stress_pairs.py 12 2>&1 | tail -1 ) >> $O/stress_pairs.txt
fi
ls -la $O
