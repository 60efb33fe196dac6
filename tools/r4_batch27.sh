#!/bin/bash
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests_b27.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/gpu_tests_b27.log
python tools/host_profile_joint.py > gpurun_out/host_profile_joint.txt 2>&1; grep -E "host issue|TRAIN step" gpurun_out/host_profile_joint.txt
