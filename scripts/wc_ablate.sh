# k_scatter1_wc alone on one c3 batch (3.67 M reads) under the measurement switches (temporary wiring of BFCG_ABLATE into the variant)
cd $GRAFT_REPO_ROOT
for cfg in ${CFGS:-"0 0" "1 0" "1 2048" "1 1280" "1 3328"}; do
  set -- $(echo $cfg | tr : " ")
  if [ "$1" = "1" ]; then export BFCG_S1_WC=1; else unset BFCG_S1_WC; fi
  if [ "$2" != "0" ]; then export BFCG_ABLATE=$2; else unset BFCG_ABLATE; fi
  echo "== BFCG_S1_WC=$1 BFCG_ABLATE=$2"; timeout 60 python scripts/s1_ablate.py 2>&1 | grep "scatter1 ms" | tail -2
done
