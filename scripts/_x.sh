export TMPDIR=/tmp; mkdir -p gpurun_out/r02e
timeout 600 python scripts/r02_lanes.py > gpurun_out/r02e/lanes_q4.log 2>&1; cat gpurun_out/r02e/lanes_q4.log
GPU_MAX_HW_QUEUES=8 timeout 600 python scripts/r02_lanes.py > gpurun_out/r02e/lanes_q8.log 2>&1; cat gpurun_out/r02e/lanes_q8.log
