for i in 1 2; do for t in 384 512 256 768; do
r=$(CSSHIP_BGEMM_TARGET_WGS=$t timeout -k 10 200 python bench.py --mode train --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('%.4f' % d['ms_per_step'])")
echo "target_wgs=$t train ms/step $r"; done; done
