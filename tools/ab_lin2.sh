for rep in 1 2; do for v in 0 1; do echo "ACAI_LIN2_WFIRST=$v"; ACAI_LIN2_WFIRST=$v python bench.py --legs "" --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done; done
