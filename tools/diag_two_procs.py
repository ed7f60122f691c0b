"""Diagnosis: two processes on ONE GPU, the same deterministic forward repeated -- do the bits ever differ?  (round 4: the two-process
RCCL refusal test once saw a forward differ from the one before it.)   python tools/diag_two_procs.py [child rank]"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) > 1:
    rank = int(sys.argv[1])
    import numpy as np
    import __graft_entry__ as graft
    graft.package()
    from dream_go_amd import api, synth
    net = api.Network.from_json(synth.to_json(synth.make_tensors(2, 64, seed=700 + rank)))
    features = synth.make_features(4, seed=701)
    outs = []
    for rep in range(12):
        with net.get_workspace(4) as ws:
            v, p = api.forward(ws, features).unwrap()
            outs.append((v.view(np.uint16).copy(), p.view(np.uint16).copy(), ws.conv_kernel()))
        if rep == 5 and os.environ.get("DIAG_RCCL"):
            try:
                api.Comm.create(0, 2, rank, open(sys.argv[2], "rb").read())
            except api.Error as e:
                print("rank", rank, "comm create ->", e.status, flush=True)
        time.sleep(0.01 * (rank + 1))
    base = outs[0]
    report = []
    for i, o in enumerate(outs):
        dv = int((o[0] != base[0]).sum())
        dp = int((o[1] != base[1]).sum())
        report.append((i, dv, dp))
    print("RESULT", json.dumps({"rank": rank, "kernel": outs[0][2], "diffs_vs_first": report}), flush=True)
    raise SystemExit(0)

env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
for mode in ("plain", "rccl"):
    ident = "/tmp/diag_ident"
    if mode == "rccl":
        sys.path.insert(0, ROOT)
        import __graft_entry__ as graft
        graft.package()
        from dream_go_amd import api
        open(ident, "wb").write(api.Comm.unique_id())
        env["DIAG_RCCL"] = "1"
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), str(r), ident], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env) for r in (0, 1)]
    for p in procs:
        out, _ = p.communicate(timeout=200)
        print(mode, [ln for ln in out.splitlines() if "RESULT" in ln or "comm create" in ln])
