for a in 0 1 2 3; do echo "ABL=$a"; WDNO_TB_ABLATE=$a timeout 100 python tools/bench_tattn.py 2>&1 | grep "with gradients" | cut -c1-120; done
