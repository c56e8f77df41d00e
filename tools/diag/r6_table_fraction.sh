#!/bin/bash
# the VALU / LDS balance point of the headline kernel -- ONE sweep.  The first radix-4 step of every forward
# transform takes its products from an LDS table for k of its 4 register groups and multiplies for the rest (ntt_wave.hpp
# BR_TAB_GROUPS); k = 4 is the default build, k = 0 equals "br_digit_table" 0.
#   here:   bash peba1_amd/csrc/build.sh && bash tools/diag/build_variants.sh tg0="-DBR_TAB_GROUPS=0" tg1="-DBR_TAB_GROUPS=1" tg2="-DBR_TAB_GROUPS=2" tg3="-DBR_TAB_GROUPS=3"
#   GPU:    gpurun --timeout 1200 -- 'bash tools/diag/r6_table_fraction.sh'   ->  gpurun_out/r06_tabfrac/ab.txt + summary on stdout
# Adopt a middle point only if it wins by >= 1.5 % at G = 4,096 AND G = 512; else record and close (profiles/r06_ab_table_fraction.txt).
exec bash tools/diag/ab.sh r06_tabfrac "default tg3 tg2 tg1 tg0" "256 512 4096 4096"
