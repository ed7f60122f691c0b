#!/bin/sh
# CPU a host thread burns while it waits for the device, per wait mode (engine.cpp: wait_submission).
for m in poll spin event; do echo "DGNN_WAIT=$m"; DGNN_WAIT=$m python tools/sync_cpu_probe.py 2>&1 | tail -1; done
