for d in 0 1 2 4 8 16 32 3 6 14 30 31 63; do echo "DBG=$d"; DPF_FLOW_DBG=$d python tools/flow_sweep.py 2>/dev/null | grep "L=14" | grep -E "bf16 |bf16x3"; done
