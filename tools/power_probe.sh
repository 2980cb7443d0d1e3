# Developer tool (GPU box): socket power and shader clock while a command runs.   bash tools/power_probe.sh <command...>
"$@" > gpurun_out/power_probe_cmd.log 2>&1 &
pid=$!
sleep 2
while kill -0 $pid 2>/dev/null; do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -i "Socket Graphics Package Power\|sclk clock level\|Average Graphics Package Power" | tr '\n' ' '; echo
  sleep 0.5
done
wait $pid
tail -2 gpurun_out/power_probe_cmd.log
