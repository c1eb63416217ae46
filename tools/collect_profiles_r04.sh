#!/bin/bash
# Everything profiles/r04_* is made of, in one go on the GPU box:  gpurun -- 'bash tools/collect_profiles_r04.sh'
# Output lands in gpurun_out/prof4/ (copy what should be judged into profiles/).  Every step runs under its own timeout.
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/prof4
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
py() { timeout ${T:-300} python "$@"; }
T=400 py $R/bench.py --steps 20 --warmup 5 > $O/bench_20steps.json 2> $O/bench_20steps.err
py $R/bench.py --steps 120 --warmup 6 --no-cpu-baseline --no-secondary > $O/bench_120steps.json 2>/dev/null
py $R/bench.py --steps 60 --warmup 6 --scene clutter --no-cpu-baseline --no-secondary > $O/bench_clutter.json 2>/dev/null
py $R/bench.py --config extract --steps 60 --warmup 6 --no-cpu-baseline --no-secondary > $O/bench_extract.json 2>/dev/null
py $R/bench.py --steps 60 --warmup 6 --include-h2d --no-cpu-baseline --no-secondary > $O/bench_include_h2d.json 2>/dev/null
T=400 py $R/bench.py --config dense128 --steps 20 --warmup 3 > $O/bench_dense128.json 2>/dev/null
# kernel trace of the default bench
rm -rf /tmp/kb; timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/kb -o kb -- python $R/bench.py --steps 60 --warmup 6 --no-cpu-baseline --no-secondary > $O/bench_under_rocprof.json 2>/dev/null
python $R/tools/prof_summary.py /tmp/kb/kb_results.db "rocprofv3 --kernel-trace --stats -- python bench.py --steps 60 --warmup 6 --no-cpu-baseline --no-secondary" > $O/kernel_stats_bench.txt 2>&1
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
# stage 1 and the match kernels: SQ counters
timeout 400 bash $R/tools/pmc_stage1x.sh $O/pmc_stage1x.txt > /dev/null 2>&1
( timeout 300 bash $R/tools/pmc_kernel.sh k_match_screen python $R/tools/match_time.py; echo "-- k_match_prep"; timeout 300 bash $R/tools/pmc_kernel.sh k_match_prep python $R/tools/match_time.py ) > $O/pmc_match.txt 2>&1
# per-kernel table of the batched launches
timeout 900 bash $R/tools/collect_roofline_table.sh > /dev/null 2>&1; cp $R/gpurun_out/prof/roofline_table_batched.txt $O/roofline_table_batched.txt
( py $R/tools/enc_table.py 2>&1 | tail -2; echo "the exact-f32 stage 1 (caelo_set_encoder_reference; the tool reads CAELO_ENC_S1=f32):"; CAELO_ENC_S1=f32 py $R/tools/enc_table.py 2>&1 | tail -2 ) > $O/enc_table.txt
( py $R/tools/enc_layer_errors.py 2>&1 | tail -4; echo "the exact-f32 stage 1:"; CAELO_ENC_S1=f32 py $R/tools/enc_layer_errors.py 2>&1 | tail -4 ) > $O/layer_errors.txt
py $R/tools/stage1_density_sweep.py > $O/stage1_density_sweep.txt 2>&1
py $R/tools/match_stats.py 2>&1 | tail -4 > $O/match_stats.txt
py $R/tools/upload_overlap_check.py 2>&1 | tail -8 > $O/upload_overlap.txt
# the front half: SQ counters per kernel, and the round-2 forms of the response layer / key point selection beside the defaults
timeout 600 bash $R/tools/pmc_front.sh > $O/pmc_front.txt 2>&1
bash $R/tools/front_times.sh > $O/front_times.txt 2>&1
( cd $R && T=600 py cae-lo_amd/run_sequence.py --synthetic 4541 --pool 49 --quantum 0.001 --chunk 240 --out $O/poses_kitti00_sized.txt 2>&1 | tail -2 ) > $O/run_sequence_4541.txt
( cd $R && timeout 300 python tools/stress_pairs.py 12 2>&1 | tail -1 ) > $O/stress_pairs.txt
( cd $R && bash tools/s1x_sweep.sh ) > $O/stage1_slots.txt 2>&1
( cd $R && T=1500 py tools/parity_soak.py --frames 200 --out $O/parity_soak.txt ) > $O/parity_soak.log 2>&1
rm -f $O/poses_kitti00_sized.txt
ls -la $O
