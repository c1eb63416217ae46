#!/bin/bash
# Everything profiles/r06_* is made of (bench legs, kernel trace + timeline, roofline launches, live PMC, file-mode run_sequence):
#   gpurun -- 'bash tools/collect_profiles_r06.sh'      -> gpurun_out/prof6/ (copy what should be judged into profiles/)
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/prof6
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
py() { timeout ${T:-300} python "$@"; }
T=700 py $R/bench.py --steps 20 --warmup 5 > $O/bench_20steps.json 2> $O/bench_20steps.err
py $R/bench.py --steps 120 --warmup 6 --no-cpu-baseline --no-secondary --no-pmc > $O/bench_120steps.json 2>/dev/null
py $R/bench.py --steps 120 --warmup 6 --no-cpu-baseline --no-secondary --no-pmc --no-certify > $O/bench_120steps_no_certify.json 2>/dev/null
py $R/bench.py --steps 60 --warmup 6 --scene clutter --no-cpu-baseline --no-secondary --no-pmc > $O/bench_clutter.json 2>/dev/null
py $R/bench.py --steps 60 --warmup 6 --include-h2d --no-cpu-baseline --no-secondary --no-pmc > $O/bench_include_h2d.json 2>/dev/null
rm -rf /tmp/kb; timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/kb -o kb -- python $R/bench.py --steps 60 --warmup 6 --no-cpu-baseline --no-secondary --no-pmc > $O/bench_under_rocprof.json 2>/dev/null
python $R/tools/prof_summary.py /tmp/kb/kb_results.db "rocprofv3 --kernel-trace --stats -- python bench.py --steps 60 --warmup 6 --no-cpu-baseline --no-secondary --no-pmc" > $O/kernel_stats_bench.txt 2>&1
python $R/tools/timeline.py /tmp/kb/kb_results.db 1700 400 > $O/timeline_bench.txt 2>&1
rm -rf /tmp/rl; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/rl -o rl -- python $R/tools/roofline_launch.py 30 8 match > /dev/null 2>&1
python $R/tools/prof_summary.py /tmp/rl/rl_results.db "rocprofv3 --kernel-trace --stats -- python tools/roofline_launch.py 30 8 match   (8 frames = 24576 patches per encoder launch, 8 pairs per match launch)" > $O/kernel_stats_roofline_launch.txt 2>&1
rm -rf /tmp/rp; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/rp -o rp -- python $R/tools/roofline_launch.py 30 8 plain > /dev/null 2>&1
python $R/tools/prof_summary.py /tmp/rp/rp_results.db "rocprofv3 --kernel-trace --stats -- python tools/roofline_launch.py 30 8 plain   (the same launches through caelo_encode: the production instantiations, back to back)" > $O/kernel_stats_roofline_launch_plain.txt 2>&1
T=400 py $R/tools/pmc_live.py $O/pmc_live.json > /dev/null 2>&1
( cd $R && T=600 py cae-lo_amd/run_sequence.py --synthetic 4541 --pool 49 --quantum 0.001 --chunk 240 --out $O/poses_kitti00_sized.txt 2>&1 | tail -2 ) > $O/run_sequence_4541.txt
( THREADS="16 16" timeout 900 bash $R/tools/run_sequence_files_probe.sh 4541 2>&1 | grep -v amdgpu.ids | grep -E "wrote|frames/s|host seconds|loader threads" ) > $O/run_sequence_files_4541.txt
rm -f $O/poses_kitti00_sized.txt
ls -la $O
# registers / LDS / scratch of every kernel (a device function that stops being inlined shows here first: DESIGN.md 6)
( cd $R && for f in ring voxel encoder match icp export frame pipeline extend config5 dedup kdorder certify seqload; do bash tools/kernel_resources.sh $f.hip 2>/dev/null | grep -v rocprim; done ) > $O/kernel_resources.txt 2>&1
( cd $R && PROBE_LANES=1,4,8 timeout 300 python tools/ties_many_probe.py 64 96 2>&1 | grep -v amdgpu.ids | grep -E "tied|lanes|match_pose" ) > $O/ties_many_probe.txt
