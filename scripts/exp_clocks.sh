# engine clock / power while the step replays (rocm-smi sampled beside a long bench run)
python bench.py --steps 4000 --warmup 20 --no-cpu-baseline --no-train > /tmp/b.json 2>/dev/null &
BP=$!
for i in $(seq 1 30); do sleep 2; echo -n "t=$((2*i))s "; rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Package Power" | sed 's/GPU\[0\]\s*: //' | tr '\n' ' '; echo; done
wait $BP
cut -c1-200 /tmp/b.json
