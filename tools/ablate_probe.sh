for a in 0 1 2 3 7; do echo "ABLATE=$a"; ARB_ABLATE=$a timeout 100 python tools/gs_stats_probe.py 0 2>&1 | grep "^  0" | cut -c60-200; done
