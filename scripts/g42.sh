#!/bin/bash
cd "$GRAFT_REPO_ROOT"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -pthread scripts/micro/launch_rate.hip -o /tmp/launch_rate 2>&1 | grep -v warning | head -5
for q in 4 8; do echo "GPU_MAX_HW_QUEUES=$q"; GPU_MAX_HW_QUEUES=$q /tmp/launch_rate 18000; done
