for cfg in "8 2" "16 2" "8 1" "16 1" "12 3" "24 2"; do set -- $cfg
python bench.py --steps 5 --warmup 2 --cpu-sample 0 --selfplay-leaves $1 --selfplay-batches $2 --selfplay-moves 8 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['selfplay']; print('leaves $1 batches $2:', {k: round(s[k],2) for k in ('evals_per_sec','mean_device_batch','games_per_hour','seconds','cache_hit_ratio','evals_per_move')})"
done
