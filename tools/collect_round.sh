#!/bin/bash
# Round-end collection on one box (run through gpurun from the repo root): GPU tests, tools/profile_round.sh (kernel stats + PMC passes), frame table, one-shot
# kernel table, library-convolution list.  The PMC summary is then copied to profiles/ and tools/pmc_traffic_summary.py run locally; the bench lines follow in a
# second call so that bench.py finds the traffic file of the same sources.
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -4 > gpurun_out/r06_gpu_tests_final.txt
tools/profile_round.sh r06 > gpurun_out/r06_profile_round.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/frame_layers.py --single > gpurun_out/r06_frame_layers.txt 2>&1
python tools/profile_oneshot_ops.py > gpurun_out/r06_oneshot_ops.txt 2>&1
python tools/list_library_convs.py --oneshot > gpurun_out/r06_library_convs_oneshot.txt 2>&1
cat gpurun_out/r06_gpu_tests_final.txt; tail -3 gpurun_out/r06_profile_round.log
