#!/bin/bash
# usage: bash tools/micro/thin_ablate.sh <prefix> <variants...>   (libs tools/micro/ablate_libs/lib<prefix>_<v>.so)
PRE=$1; shift
for v in "$@"; do
  VEGANS_ABLATE_LIB=$(pwd)/tools/micro/ablate_libs/lib${PRE}_$v.so timeout -k 10 120 python tools/micro/thin_ablate.py || exit 1
done
