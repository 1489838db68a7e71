python -m pytest tests/test_gpu_parity.py -q -m gpu -k "resize or chain" -x 2>&1 | tail -2
for a in "3840 2160 16 1280 720" "3840 2160 16 1706 960" "1920 1080 16 1280 720" "1920 1080 8 3840 2160" "3840 2160 1 1280 720"; do python tools/bench_chain_geom.py $a 2>&1 | tail -1; done
python tools/bench_resize.py 2>&1 | tail -6
