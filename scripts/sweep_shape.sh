for cfg in "4 5" "2 10" "2 5" "3 7" "5 4" "4 3" "6 4" "8 3" "3 4" "4 5"; do
  set -- $cfg
  python bench.py --steps 20 --warmup 5 --no-diagnostics --streams $1 --frames-per-launch $2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('S=$1 G=$2', d['value'], d['ms_per_step'])"
done
