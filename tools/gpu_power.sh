#!/bin/bash
# is the C5 / C2 workload power-limited?  package power + shader clock sampled from sysfs while bench.py runs
TAG=${1:-power}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
ls /sys/class/drm/ > $OUT/sysfs.txt 2>&1; ls /sys/class/drm/card*/device/hwmon/hwmon*/ >> $OUT/sysfs.txt 2>&1
rocm-smi --showpower --showclocks --showmaxpower > $OUT/smi_idle.txt 2>&1
for m in 0 auto; do
  if [ $m = auto ]; then unset NOMAD_BF16_N192; else export NOMAD_BF16_N192=$m; fi
  timeout 300 python3 tools/power_sample.py -- python3 bench.py --dtype bf16 --seconds 30 --batch 32 --refs 4 --steps 150 --warmup 3 --no-cpu-baseline --no-profile > $OUT/c5_$m.txt 2> $OUT/c5_$m.err; echo "c5 $m exit $?"
  tail -n 1 $OUT/c5_$m.txt | cut -c1-1200
  grep -o '"value": [0-9.]*' $OUT/c5_$m.txt | head -1
done
unset NOMAD_BF16_N192
timeout 300 python3 tools/power_sample.py -- python3 bench.py --steps 30 --warmup 2 --no-cpu-baseline --no-profile --no-also > $OUT/c2.txt 2> $OUT/c2.err; echo "c2 exit $?"
tail -n 1 $OUT/c2.txt | cut -c1-1200
grep -o '"value": [0-9.]*' $OUT/c2.txt | head -1
(rocm-smi --showpower --showclocks > $OUT/smi_after.txt 2>&1)
head -30 $OUT/smi_idle.txt
