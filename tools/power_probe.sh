# Package power and shader clock while a workload loops (is it running at the chip's power budget?): bash tools/power_probe.sh "<command>" [seconds]
cmd=$1; secs=${2:-8}
( eval "$cmd" 2>/dev/null | tail -1 ) &
wl=$!
sleep ${3:-6}
for i in $(seq 1 12); do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power \(W\)|sclk" | sed 's/.*: //' | tr '\n' ' '; echo; sleep 0.4; done
wait $wl
echo "-- idle:"; sleep 1; rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power \(W\)|sclk" | sed 's/.*: //' | tr '\n' ' '; echo
rocm-smi --showmaxpower 2>/dev/null | grep -i "max" | head -2
