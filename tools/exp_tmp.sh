timeout -k 10 600 python tools/eps_probe_tmp.py 2>&1 | tail -14
