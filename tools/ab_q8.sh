#!/bin/bash
# As ab_quick.sh with 8 hardware queues per process instead of HIP's default of 4.
export GPU_MAX_HW_QUEUES=8
exec bash "$(dirname "$0")/ab_quick.sh" "$@"
