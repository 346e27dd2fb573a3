#!/bin/bash
# Runs ON the GPU box: A/B of library builds on the same box.  usage: tools/ll_ab.sh "<lib suffixes>" "<tune cfgs...>"
# (lib suffix "" = libmgym.so; "_old" = libmgym_old.so ...).  LL_TUNE_ENVS selects the population.
L=modurl_gym_amd
cp $L/libmgym.so /tmp/libmgym_cur.so
libs="$1"; shift
for s in $libs; do
  [ "$s" = "cur" ] && cp /tmp/libmgym_cur.so $L/libmgym.so || cp $L/libmgym_$s.so $L/libmgym.so
  echo "== lib $s (envs ${LL_TUNE_ENVS:-default})"
  tools/ll_tune.sh "$@"
done
cp /tmp/libmgym_cur.so $L/libmgym.so
