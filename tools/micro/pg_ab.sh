# Same-box comparison of the bf16 contractions across builds of the library: the product and every
# tools/micro/ablate_libs/libpg_*.so (builds of other source states, see pgemm_ablate.py).  usage: bash tools/micro/pg_ab.sh
set -e
R=$(pwd)
timeout -k 10 120 python tools/micro/pgemm_ablate.py
for l in tools/micro/ablate_libs/libpg_*.so; do
  VEGANS_ABLATE_LIB=$R/$l timeout -k 10 120 python tools/micro/pgemm_ablate.py
done
timeout -k 10 120 python tools/micro/pgemm_ablate.py
