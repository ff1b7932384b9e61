# A/B of cut-kernel builds in one call:  bash tools/r03_cut_ab.sh "<tag> ..."   (tags under proton_amd/lib/variants/)
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do
  for T in $1; do
    echo -n "$T: "
    PA_LIB=$PWD/proton_amd/lib/variants/$T/libproton_amd.so timeout -k 10 200 python tools/r03_cut_clock.py 512 2 2>&1 | grep "cut kernel"
  done
done
for T in $1; do
  echo -n "$T: "
  PA_LIB=$PWD/proton_amd/lib/variants/$T/libproton_amd.so PA_CUT_CLOCK=1 timeout -k 10 200 python tools/r03_cut_clock.py 512 2 2>&1 | grep -m1 "PA_CUT_CLOCK"
done
