set -e
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/clk; cd $GRAFT_REPO_ROOT/gpurun_out/clk
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 $GRAFT_REPO_ROOT/profiles/r01/mfma_f64_clocks.hip -o /tmp/loop
(rocm-smi --showclocks --showpower 2>&1 | grep -i "sclk\|power\|fclk\|mclk" | head -8) > idle.txt || true
/tmp/loop > loop.txt &
PID=$!
sleep 2
for i in 1 2 3; do (rocm-smi --showclocks --showpower 2>&1 | grep -i "sclk\|power" | head -6) >> busy.txt || true; sleep 1; done
wait $PID
