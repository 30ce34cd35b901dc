#!/bin/bash
# usage: tools/exp_clocks.sh <clips> <steps> : sample the GPU clocks / power while bench.py runs a long timed loop
CLIPS=${1:-4096}; STEPS=${2:-1500}
F=$(ls /sys/class/drm/card*/device/hwmon/hwmon*/freq1_input 2>/dev/null | head -1)
P=$(ls /sys/class/drm/card*/device/hwmon/hwmon*/power1_average /sys/class/drm/card*/device/hwmon/hwmon*/power1_input 2>/dev/null | head -1)
( for i in $(seq 1 200); do echo "$(date +%s.%N | cut -c1-14) sclk_hz=$(cat $F 2>/dev/null) power_uW=$(cat $P 2>/dev/null)"; sleep 0.1; done ) > /tmp/clk.log &
SP=$!
timeout 300 python bench.py --cpu-seconds 0 --e2e-clips 0 --clips $CLIPS --steps $STEPS --warmup 8 2>/dev/null | python tools/brief.py
kill $SP 2>/dev/null
echo "power cap: $(cat $(dirname $P)/power1_cap 2>/dev/null) uW; temps: $(cat $(dirname $P)/temp*_input 2>/dev/null | tr '\n' ' ')"
awk '{split($2,a,"="); split($3,b,"="); if (a[2] > 2000000000) {n++; s+=a[2]; p+=b[2]; if (a[2]>mx) mx=a[2]}} END {if (n) printf "samples %d  mean sclk %.0f MHz  max %.0f MHz  mean power %.0f W\n", n, s/n/1e6, mx/1e6, p/n/1e6}' /tmp/clk.log
sort -t= -k2 -n /tmp/clk.log | awk '{print $2}' | uniq -c | sort -rn | head -5
