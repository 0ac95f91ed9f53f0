#!/bin/bash
# poll GPU clocks/power every ~0.25 s into $1 until the file $1.stop appears
out=$1
while [ ! -e $out.stop ]; do
  echo "$(date +%s.%N) $(cat /sys/class/drm/card*/device/pp_dpm_sclk 2>/dev/null | grep '\*' | tr '\n' ' ') | $(cat /sys/class/drm/card*/device/pp_dpm_mclk 2>/dev/null | grep '\*' | tr '\n' ' ') | $(cat /sys/class/drm/card*/device/hwmon/hwmon*/power1_average 2>/dev/null | tr '\n' ' ') | $(cat /sys/class/drm/card*/device/pp_dpm_fclk 2>/dev/null | grep '\*' | tr '\n' ' ')" >> $out
  sleep 0.25
done
