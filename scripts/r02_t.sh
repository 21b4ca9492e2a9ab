#!/bin/bash
for a in "--streams 2" "--streams 4" "--streams 3" "--streams 4 --frames-per-launch 4"; do
timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-diagnostics --steps 256 --warmup 32 $a | python3 -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c3 $a', r['value'], r['ms_per_step'], r['roofline']['kernel_ms_avg'])"
done
timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-diagnostics --steps 20 --warmup 5 --streams 4| python3 -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c3 driver shape S=4', r['value'], r['config']['frames_per_launch'])"
timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-diagnostics --steps 20 --warmup 5 --streams 2| python3 -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c3 driver shape S=2', r['value'], r['config']['frames_per_launch'])"
