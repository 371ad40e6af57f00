#!/bin/bash
# what does this box expose for socket power / shader clock?  (bench.py's PowerSampler reads the first that exists)
for d in /sys/class/drm/card*/device; do
  echo "== $d vendor $(cat $d/vendor 2>/dev/null)"
  ls $d/hwmon/*/ 2>/dev/null | tr '\n' ' '; echo
  for f in $d/hwmon/*/power1_average $d/hwmon/*/power1_input $d/hwmon/*/freq1_input $d/hwmon/*/freq1_label $d/hwmon/*/power1_cap; do
    [ -e $f ] && echo "$f = $(cat $f 2>&1)"
  done
  [ -e $d/pp_dpm_sclk ] && { echo pp_dpm_sclk:; cat $d/pp_dpm_sclk; }
done
python3 -c "import amdsmi; print('amdsmi module ok')" 2>&1 | tail -1
which amd-smi rocm-smi
rocm-smi --showpower --showclocks 2>&1 | head -30
