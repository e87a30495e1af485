for d in 16 47 79 111 64 32; do echo -n "dbg=$d "; RLREP_NCDX_DBG=$d python tools/stage_times.py 2>&1 | grep -E "noise critic dX"; done
