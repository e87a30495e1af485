python tools/stage_times.py 2>&1 | grep -E "noise critic|train\(\)|sum of"
timeout 300 python -m pytest tests/test_hip_parity.py -m gpu -q -k vlsac 2>&1 | tail -2
