#!/bin/bash
# one process per configuration (see vmm_probe4.hip)
P=tools/vmm_probe4
for rep in 1 2; do
$P malloc; $P malloc 0 0 40; $P malloc 0 0 100
$P 0; $P 0 0 0 40
$P 1024; $P 1024 1; $P 1024 0 32; $P 1024 0 192; $P 1024 0 0 40; $P 1024 1 0 100
$P 256; $P 256 1
$P 64; $P 64 1
$P 32 1
done
$P 2; $P 2 1
