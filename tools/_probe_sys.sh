ls /sys/class/drm/ 2>&1 | head -20
for d in /sys/class/drm/card*/device; do echo "== $d"; ls $d | tr '\n' ' ' | head -c 1500; echo; ls $d/hwmon/*/ 2>/dev/null | tr '\n' ' '; echo; done
for f in /sys/class/drm/card*/device/hwmon/hwmon*/{power1_average,power1_input,freq1_input,freq1_label,power1_cap}; do echo "$f: $(cat $f 2>&1)"; done
for f in /sys/class/drm/card*/device/pp_dpm_sclk; do echo $f; cat $f 2>&1 | head -5; done
time rocm-smi --showclocks --showpower 2>&1 | tail -12
python3 -c "import amdsmi; print('amdsmi ok')" 2>&1 | tail -1
nproc; cat /sys/fs/cgroup/cpu.max
