#!/bin/bash
# Timing-only ablations of k_distance_panel (the staged pass alone at C3, scratch/dist_loop.py; wrong results by design):
# which parts of the launch add up and which overlap.  Variants are built with scratch/build_variant.py <name> -D... --no-audit.
for L in shipped lib_nostore.so lib_noepi.so lib_nostream.so lib_nopanel.so lib_noepi_nostream.so lib_noepi_nopanel.so lib_mfmaonly.so; do
  python scratch/dist_loop.py $L 16 2>&1 | grep "distance pass"
done
