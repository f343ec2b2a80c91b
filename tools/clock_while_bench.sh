# samples rocm-smi's clocks and power while bench.py's timed steps run (is the matrix pipe's ceiling a clock/power ceiling?)
python3 bench.py --no-extras --no-cpu-baseline --no-emulation --steps 40 --warmup 3 > gpurun_out/bench_clk.json 2> gpurun_out/bench_clk.err &
BP=$!
sleep 8
for i in $(seq 1 40); do rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|power\|mclk" | tr '\n' ' '; echo; sleep 0.2; done > gpurun_out/clk_samples.txt
wait $BP
tail -25 gpurun_out/clk_samples.txt
