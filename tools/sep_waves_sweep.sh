for w in 64 128 256 512 1024 2048; do echo "== waves $w"; SGE_SEPARATION_WAVES=$w timeout -k 10 120 python tools/separation_bench.py 8192 2>&1 | grep "^footprint" | cut -c1-120; done
