#!/bin/bash
# cgroup CPU throttling around a command: prints the nr_throttled / throttled_usec deltas
s() { grep -E "nr_throttled|throttled_usec" /sys/fs/cgroup/cpu.stat | tr '\n' ' '; }
echo "before: $(s)"
"$@"
echo "after:  $(s)"
