# Diagnostic: CFS throttling (cgroup cpu.max) of a command: prints nr_throttled / throttled_usec deltas.
s0=$(grep -E "nr_throttled|throttled_usec|usage_usec" /sys/fs/cgroup/cpu.stat | tr '\n' ' ')
"$@"
s1=$(grep -E "nr_throttled|throttled_usec|usage_usec" /sys/fs/cgroup/cpu.stat | tr '\n' ' ')
echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max)   before: $s0   after: $s1"
