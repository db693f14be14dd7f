#!/bin/bash
# Diagnostic: sample rocm-smi clocks / power every 0.5 s while a command runs.  usage: scripts/sample_clocks.sh <cmd...>
"$@" > /tmp/cmd_out.txt 2>&1 &
PID=$!
while kill -0 $PID 2>/dev/null; do
  /opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|mclk" | tr '\n' ' '; echo
  sleep 0.5
done
tail -3 /tmp/cmd_out.txt
