for r in 1 2; do for v in 4 1 2; do for m in 13B 65B; do
MI355_ATTN_SPLITS=$v python bench.py --model $m --steps 32 --no-cpu-baseline --no-tp 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('round $r splits=$v $m', d['value'], d['ms_per_step'])"
done; done; done
for v in 4 1 2; do MI355_ATTN_SPLITS=$v python bench.py --model 65B --prompt-len 1900 --steps 32 --warmup 8 --no-cpu-baseline --no-tp 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('long context (1900) splits=$v 65B', d['value'], d['ms_per_step'])"; done
