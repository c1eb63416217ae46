#!/bin/bash
# Everything profiles/r05_* is made of, in one go on the GPU box:  gpurun -- 'bash tools/collect_profiles_r05.sh'
# Output lands in gpurun_out/prof5/ (copy what should be judged into profiles/).  Every step runs under its own timeout.
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/prof5
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
py() { timeout ${T:-300} python "$@"; }
T=600 py $R/bench.py --steps 20 --warmup 5 > $O/bench_20steps.json 2> $O/bench_20steps.err
py $R/bench.py --steps 120 --warmup 6 --no-cpu-baseline --no-secondary --no-pmc > $O/bench_120steps.json 2>/dev/null
py $R/bench.py --steps 120 --warmup 6 --no-cpu-baseline --no-secondary --no-pmc --no-certify > $O/bench_120steps_no_certify.json 2>/dev/null
py $R/bench.py --steps 60 --warmup 6 --scene clutter --no-cpu-baseline --no-secondary --no-pmc > $O/bench_clutter.json 2>/dev/null
py $R/bench.py --steps 60 --warmup 6 --include-h2d --no-cpu-baseline --no-secondary --no-pmc > $O/bench_include_h2d.json 2>/dev/null
# kernel trace of the default bench
rm -rf /tmp/kb; timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/kb -o kb -- python $R/bench.py --steps 60 --warmup 6 --no-cpu-baseline --no-secondary --no-pmc > $O/bench_under_rocprof.json 2>/dev/null
python $R/tools/prof_summary.py /tmp/kb/kb_results.db "rocprofv3 --kernel-trace --stats -- python bench.py --steps 60 --warmup 6 --no-cpu-baseline --no-secondary --no-pmc" > $O/kernel_stats_bench.txt 2>&1
python $R/tools/timeline.py /tmp/kb/kb_results.db 1700 400 > $O/timeline_bench.txt 2>&1
# the launches bench.py's roofline objects time: 8 frames per encoder launch, 8 pairs per match launch
rm -rf /tmp/rl; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/rl -o rl -- python $R/tools/roofline_launch.py 30 8 match > /dev/null 2>&1
python $R/tools/prof_summary.py /tmp/rl/rl_results.db "rocprofv3 --kernel-trace --stats -- python tools/roofline_launch.py 30 8 match   (8 frames = 24576 patches per encoder launch, 8 pairs per match launch)" > $O/kernel_stats_roofline_launch.txt 2>&1
T=400 py $R/tools/pmc_live.py $O/pmc_live.json > /dev/null 2>&1
# per-kernel table of the batched launches, one stream
timeout 900 bash $R/tools/collect_roofline_table.sh > /dev/null 2>&1; cp $R/gpurun_out/prof/roofline_table_batched.txt $O/roofline_table_batched.txt
( py $R/tools/enc_table.py 2>&1 | tail -2 ) > $O/enc_table.txt
( py $R/tools/enc_layer_errors.py 2>&1 | tail -4 ) > $O/layer_errors.txt
( py $R/tools/cert_cost_probe.py 120 2>&1 | grep -v amdgpu.ids; py $R/tools/cert_cost_probe.py 20 2>&1 | grep -v amdgpu.ids ) > $O/cert_cost.txt
( cd $R && T=600 py cae-lo_amd/run_sequence.py --synthetic 4541 --pool 49 --quantum 0.001 --chunk 240 --out $O/poses_kitti00_sized.txt 2>&1 | tail -2 ) > $O/run_sequence_4541.txt
( cd $R && timeout 300 python tools/stress_pairs.py 12 2>&1 | tail -1 ) > $O/stress_pairs.txt
( THREADS="32 32" timeout 900 bash $R/tools/run_sequence_files_probe.sh 4541 2>&1 | grep -v amdgpu.ids | grep -E "wrote|frames/s|host seconds" ) > $O/run_sequence_files_4541.txt
rm -f $O/poses_kitti00_sized.txt
# what the upload mode loses and why: unrelated H2D copies beside the resident pipeline (8 x 2 MB commands vs one 16 MB command), per-stream H2D rates
( cd $R && py tools/upload_contention_probe.py 2>&1 | grep -v amdgpu.ids; py tools/h2d_stream_probe.py 2>&1 | grep -v amdgpu.ids ) > $O/upload_overlap.txt
# the tie redo (clutter scene): many frames on side streams, one frame's stages, the kd-tree build level by level (a -DKD_PROFILE build)
( cd $R && py tools/ties_many_probe.py 64 2>&1 | grep -v amdgpu.ids ) > $O/ties_many.txt
( cd $R && py tools/ties_probe.py 2>&1 | grep -v amdgpu.ids ) > $O/ties_probe.txt
[ -f $R/variants/libcaelo_kdprof.so ] && ( cd $R && CAELO_LIB=$R/variants/libcaelo_kdprof.so py tools/kd_profile_probe.py 2>&1 | grep -v amdgpu.ids ) > $O/kd_build_levels.txt
ls -la $O
