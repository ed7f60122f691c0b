#!/bin/bash
# Round 5, third GPU call: the energy table (ingredients of the K loop, capped and uncapped), then the whole -m gpu suite.
out=gpurun_out/r05c
mkdir -p $out
sh tools/mfma_energy.sh 5 mfma_energy3 > $out/mfma_energy3.txt 2>&1
grep final $out/mfma_energy3.txt | tail -21
(time python -m pytest tests -m gpu -x -q) > $out/gputest.log 2>&1
grep -E "passed|failed|error" $out/gputest.log | tail -3
python -m pytest tests/test_sharp_fixture.py tests/test_multigpu_readiness_gpu.py -m gpu -q -s 2>&1 | grep -E "logits HIP|passed|failed" | cut -c1-400
