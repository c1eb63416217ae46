#!/bin/bash
# Everything profiles/r03_* is made of, in one go on the GPU box:  gpurun -- 'bash tools/collect_profiles_r03.sh'
# Output lands in gpurun_out/prof3/ (copy what should be judged into profiles/).  Every step runs under its own timeout.
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/prof3
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
( CAELO_ENC_S1=f32 timeout 300 bash $R/tools/pmc_stage1x.sh $O/pmc_stage1_f32.txt ) > /dev/null 2>&1
# per-kernel table of the batched launches
timeout 900 bash $R/tools/collect_roofline_table.sh > /dev/null 2>&1; cp $R/gpurun_out/prof/roofline_table_batched.txt $O/roofline_table_batched.txt
( py $R/tools/enc_table.py 2>&1 | tail -2; echo "CAELO_ENC_S1=f32 (round 2's stage 1):"; CAELO_ENC_S1=f32 py $R/tools/enc_table.py 2>&1 | tail -2 ) > $O/enc_table.txt
( py $R/tools/enc_layer_errors.py 2>&1 | tail -4; echo "CAELO_ENC_S1=f32:"; CAELO_ENC_S1=f32 py $R/tools/enc_layer_errors.py 2>&1 | tail -4 ) > $O/layer_errors.txt
py $R/tools/stage1_density_sweep.py > $O/stage1_density_sweep.txt 2>&1
py $R/tools/match_stats.py 2>&1 | tail -4 > $O/match_stats.txt
py $R/tools/upload_overlap_check.py 2>&1 | tail -8 > $O/upload_overlap.txt
( for m in none wait plain host1 host2; do PROBE_MODE=$m py $R/tools/cadence_probe.py 40 10 | tail -2; done; echo '# GPU_MAX_HW_QUEUES=24:'; GPU_MAX_HW_QUEUES=24 PROBE_MODE=wait py $R/tools/cadence_probe.py 40 10 | tail -1; echo '# CAELO_PIPE_PACE=-1 (the issuing thread never waits), none / wait:'; for m in none wait; do CAELO_PIPE_PACE=-1 PROBE_MODE=$m py $R/tools/cadence_probe.py 40 10 | tail -1; done ) > $O/handover_cost.txt 2>&1
( for st in 5 10 20 40 80; do py $R/tools/host_prep_probe.py $st | tail -1; done; echo '# CAELO_PIPE_PACE=-1:'; for st in 5 10 20 40 80; do CAELO_PIPE_PACE=-1 py $R/tools/host_prep_probe.py $st | tail -1; done; echo '# a pool walked with a wrap (one failing pair in 17):'; PROBE_ORDER=wrap py $R/tools/host_prep_probe.py 20 | tail -1 ) > $O/host_pacing.txt 2>&1
py $R/tools/micro/h2d_bandwidth.py > $O/h2d_bandwidth.txt 2>&1
# the micro-benchmarks (binaries are not tracked: built here when missing)
for m in f16_mfma_subnormal:f16sub mfma_f32_order:mfma_order wave_placement:wave_placement; do
  src=$R/tools/micro/${m%%:*}.hip; bin=$R/tools/micro/${m##*:}.bin
  [ -x $bin ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -Wno-unused-result -Xclang -target-feature -Xclang -packed-fp32-ops $src -o $bin > /dev/null 2>&1
done
$R/tools/micro/f16sub.bin > $O/f16_mfma_subnormal.txt 2>&1
$R/tools/micro/mfma_order.bin > $O/mfma_f32_order.txt 2>&1
$R/tools/micro/wave_placement.bin > $O/wave_placement.txt 2>&1
# the front half: SQ counters per kernel, and the round-2 forms of the response layer / key point selection beside the defaults
timeout 600 bash $R/tools/pmc_front.sh > $O/pmc_front.txt 2>&1
( echo "# defaults"; bash $R/tools/front_times.sh; echo "# CAELO_RESPOND=valu CAELO_KP_SELECT=single (round 2's kernels)"; CAELO_RESPOND=valu CAELO_KP_SELECT=single bash $R/tools/front_times.sh "k_respond|k_kp_" ) > $O/front_times.txt 2>&1
( cd $R && T=600 py cae-lo_amd/run_sequence.py --synthetic 4541 --pool 49 --quantum 0.001 --chunk 240 --out $O/poses_kitti00_sized.txt 2>&1 | tail -1 ) > $O/run_sequence_4541.txt
( cd $R && timeout 300 python tools/stress_pairs.py 12 2>&1 | tail -1 ) > $O/stress_pairs.txt
rm -f $O/poses_kitti00_sized.txt
ls -la $O
