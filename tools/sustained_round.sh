# Developer tool (GPU box): the headline kernel SUSTAINED -- 400 back-to-back launches of 12e6 symbols (about 4.5 s of kernel time) --
# and the shader clock it holds meanwhile (GRBM_GUI_ACTIVE / 8 XCDs / dispatch duration, from one rocprofv3 pass that collects the
# counter and the kernel trace together).   bash tools/sustained_round.sh r04
TAG=${1:-r04}
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 300 python bench.py --steps 400 --warmup 5 --no-cpu-baseline > gpurun_out/${TAG}_bench_sustained.json 2> gpurun_out/bench_sustained.err
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/sust_clk
timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/sust_clk -- python3 $R/bench.py --steps 100 --warmup 2 --no-cpu-baseline > $R/gpurun_out/sust_clk.log 2>&1
cd $R
python3 - <<'PY' > gpurun_out/${TAG}_sustained_clock.txt
import csv, glob, json, os
tag = os.environ.get("TAG", "r04")
d = json.load(open(glob.glob("gpurun_out/*_bench_sustained.json")[-1]))
print("# python bench.py --steps 400 --warmup 5 --no-cpu-baseline: value %.4e symbols/s, ms_per_step %.3f, kernel_ms_avg %.3f, roofline.frac %.4f"
      % (d["value"], d["ms_per_step"], d["roofline"]["kernel_ms_avg"], d["roofline"]["frac"]))
cc = glob.glob("gpurun_out/sust_clk/*/*_counter_collection.csv")
kt = glob.glob("gpurun_out/sust_clk/*/*_kernel_trace.csv")
gui, dur = {}, {}
for row in csv.DictReader(open(cc[0])):
    if "wofdm_frames_kernel" in row["Kernel_Name"] and row["Counter_Name"] == "GRBM_GUI_ACTIVE":
        gui[row["Dispatch_Id"]] = float(row["Counter_Value"])
for row in csv.DictReader(open(kt[0])):
    if "wofdm_frames_kernel" in row["Kernel_Name"]:
        dur[row["Dispatch_Id"]] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-9
ids = sorted(set(gui) & set(dur), key=int)
clk = [gui[i] / 8.0 / dur[i] / 1e9 for i in ids]
ms = [dur[i] * 1e3 for i in ids]
n = len(ids)
print("# rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace, %d dispatches of 12e6 symbols (counter collection serialises the launches):" % n)
for name, sel in (("first 10", slice(0, 10)), ("middle 10", slice(n // 2 - 5, n // 2 + 5)), ("last 10", slice(n - 10, n))):
    print("%-10s kernel %.3f ms   clock %.3f GHz" % (name, sum(ms[sel]) / len(ms[sel]), sum(clk[sel]) / len(clk[sel])))
print("all        kernel %.3f ms (min %.3f, max %.3f)   clock %.3f GHz (min %.3f, max %.3f)"
      % (sum(ms) / n, min(ms), max(ms), sum(clk) / n, min(clk), max(clk)))
PY
cat gpurun_out/${TAG}_sustained_clock.txt
