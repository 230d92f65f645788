# clocks / power sampled by rocm-smi (its own process, started before python touches the GPU) while tools/dbg/sustain.py runs
(for i in $(seq 60); do rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "sclk|mclk|Power|Temperature \(Sensor (edge|junction)" | tr -s ' ' | tr '\n' '|'; echo; sleep 0.25; done) > gpurun_out/sustain_smi.log 2>&1 &
python3 tools/dbg/sustain.py
wait
