#!/bin/sh
# fresh 4 GB file per experiment (written by another process), on tmpfs and on the overlay fs
g++ -O2 -o /tmp/io_probe2 tools/io_probe2.cpp -lpthread || exit 1
ls /sys/devices/system/node/ | grep node; cat /sys/devices/system/node/node*/cpulist
for d in /dev/shm /tmp; do
  for cfg in "pread 8 -1" "pread 8 0" "pread 8 1" "noreuse 8 -1" "mmap 8 -1" "pread 24 -1" "mmap 24 -1"; do
    rm -f $d/probe.bin; python3 -c "
import os
b = os.urandom(1<<24)
with open('$d/probe.bin','wb') as f:
    for i in range(256): f.write(b)
"
    echo "== $d $cfg"; /tmp/io_probe2 $d/probe.bin $cfg
  done
done
rm -f /dev/shm/probe.bin /tmp/probe.bin
uname -r
