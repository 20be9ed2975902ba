export TMPDIR=/tmp
for cfg in "0 3" "1 12" "2 12" "2 6" "2 24" "0 3"; do set -- $cfg
XX_LIST_MODE=$1 XX_LIST_BLOCKS=$2 python3 tools/time_mle_eps.py 2>&1 | grep -v amdgpu.ids | head -3 | python3 -c "
import json,sys
print('list mode $1 blocks x$2:', [ (json.loads(l)['eps'], json.loads(l)['ms_per_pass']) for l in sys.stdin if l.startswith('{')])"
done
for cfg in "0 3" "2 12" "2 6" "0 3"; do set -- $cfg
XX_LIST_MODE=$1 XX_LIST_BLOCKS=$2 python3 tools/bench_configs.py --only 5 --no-lq3d --cpu-seconds 0 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('config5 list mode $1 blocks x$2:', round(d['ms_per_step'],3))"
done
XX_LIST_MODE=2 XX_LIST_BLOCKS=12 timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_mle_knobs.py -x -q -m gpu -k "gaussmle or mle" 2>&1 | tail -3
