#!/bin/bash
# The measured self-play leg of bench.py under other pool settings (leaves a search thread keeps on the device, device batches
# alive).  Prints evals/s, rollouts/s (evaluations + cache hits), the mean device batch and the CPU cores the process kept busy.
for cfg in ${CONFIGS:-"8 2" "16 2" "8 3" "4 2"}; do set -- $cfg
python bench.py --steps 5 --warmup 2 --cpu-sample 0 --selfplay-leaves $1 --selfplay-batches $2 --selfplay-moves ${MOVES:-8} --selfplay-games ${GAMES:-256} 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['selfplay']; print('leaves $1 batches $2:', {k: round(s[k],2) for k in ('host_cpus_busy','rollouts_per_sec','evals_per_sec','mean_device_batch','games_per_hour','seconds','cache_hit_ratio','evals_per_move')})"
done
