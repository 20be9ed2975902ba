python -m pytest tests/test_gpu_parity.py tests/test_gpu_surface.py -m gpu -x -q -k "identify or localize or pipeline" 2>&1 | tail -4
python tools/time_identify.py 10000 7 2>/dev/null | head -2
for c in "2048 2048" "1024 1024" "512 512" "64 64" "512 512 uint8"; do python tools/time_identify_shapes.py 7 $c 2>/dev/null | tail -1; done
