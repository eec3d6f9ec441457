for rep in 1 2; do for v in 1 0; do echo "ACAI_LN_COLSUM=$v"; ACAI_LN_COLSUM=$v python bench.py --legs mae,tf --no-cpu-baseline --steps 16 --warmup 4 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['mae']['ms_per_step'], d['tf_step']['ms_per_step'])"; done; done
