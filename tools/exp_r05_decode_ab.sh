# decode A/B experiments (round 5): bash tools/exp_r05_decode_ab.sh "<env assignments A>" "<env assignments B>" [batch ...]
A="$1"; B="$2"; shift 2
run() { env $1 python bench.py --model whisper-medium --decode --batch $2 --steps 3 --warmup 1 --no-also --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d.get('ms_per_token', d.get('config',{}).get('ms_per_token')), d['value'])"; }
for b in ${@:-64}; do for r in 1 2; do echo -n "B=$b [$A]: "; run "$A" $b; echo -n "B=$b [$B]: "; run "$B" $b; done; done
