for u in 2 3 4; do for c in 1024 512; do echo "U=$u chunk=$c"; ACAI_DATTN_U=$u ACAI_CROSS_CHUNK=$c python bench.py --legs "" --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_us'], d['roofline']['frac'])"; done; done
