cd $GRAFT_REPO_ROOT
(for i in 1 2 3 4 5 6; do timeout 60 rl-x_amd/build/mfma_peak > /dev/null; done) &
P1=$!
sleep 1.0
for i in 1 2 3; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|mclk" | head -4; sleep 0.3; done
wait $P1
echo "---- gemm probe (real tile loop)"
(for i in 1 2 3 4 5 6 7 8; do timeout 60 rl-x_amd/build/gemm_probe > /dev/null; done) &
P2=$!
sleep 1.5
for i in 1 2 3 4; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|mclk" | head -4; sleep 0.3; done
wait $P2
echo "---- idle"
rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|mclk" | head -4
