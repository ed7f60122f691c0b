#!/bin/bash
# NOTE (round 6): the switch this script toggles (pair split / head-in-tower / arrival handshake) has been removed from the sources; the script is kept as the record of how the files it names under profiles/ were produced.
# Round 5, seventh GPU call: the head convolution inside the tower launch -- the suite, then A/B on one box (test build: DGNN_HEAD_IN_TOWER).
out=gpurun_out/r05g
mkdir -p $out
(time python -m pytest tests -m gpu -q) > $out/gputest.log 2>&1
grep -E "passed|failed|error" $out/gputest.log | tail -3
grep -E "^FAILED|^ERROR" $out/gputest.log | cut -c1-200 | head
cat > /tmp/ab_head.py <<'PY'
import sys, json, time
sys.path.insert(0, ".")
import __graft_entry__ as graft
graft.package()
from dream_go_amd import api as shipped, synth
api = shipped.load_full()
import numpy as np
net = api.Network.from_json(synth.to_json(synth.make_tensors(20, 256, seed=1)))
f = synth.make_features(256, seed=2)
ws = net.get_workspace(256)
ws.upload(f)
for _ in range(5): ws.forward_resident()
ws.sync()
best = []
for rep in range(6):
    total_ms, tower_ms, launches = ws.forward_timed(60)
    best.append((256 * 60 / (total_ms * 1e-3), total_ms / 60, tower_ms / launches))
v, p = api.forward(ws, f).unwrap()
print(json.dumps({"kernel": ws.conv_kernel(), "evals_per_sec": [round(b[0]) for b in best], "ms_per_pass": round(min(b[1] for b in best), 4), "tower_ms": round(min(b[2] for b in best), 4),
                  "policy_crc": int(np.asarray(p).view(np.uint16).astype(np.uint64).sum()), "value_crc": int(np.asarray(v).view(np.uint16).astype(np.uint64).sum())}))
PY
for h in 1 0 1 0; do DGNN_HEAD_IN_TOWER=$h python /tmp/ab_head.py 2>/dev/null | sed "s/^/head_in_tower=$h /"; done
for h in 1 0; do DGNN_HEAD_IN_TOWER=$h python tools/stage_times.py --batches 192,256 --steps 60 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    print('head_in_tower=$h b', d['batch'], 'fl', d['in_flight'], 'tower %.0f headconv %.0f gemm %.0f softmax %.0f | s2a %.0f' % (d['tower'], d['head_conv'], d['policy_gemm'], d['softmax_value'], d['host']['submit_to_answers_us']), 'boards/s %.0f' % (d['batch'] * d['in_flight'] / d['host']['submit_to_answers_us'] * 1e6))
"; done
