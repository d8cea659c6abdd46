cd tools
for i in 1 2 3; do python tools_lanes.py 2 256 | grep lanes; done
for i in 1 2; do GPU_MAX_HW_QUEUES=8 python tools_lanes.py 2 256 | grep lanes; done
for i in 1 2; do GPU_MAX_HW_QUEUES=2 python tools_lanes.py 2 256 | grep lanes; done
for i in 1 2; do GPU_MAX_HW_QUEUES=1 python tools_lanes.py 2 256 | grep lanes; done
