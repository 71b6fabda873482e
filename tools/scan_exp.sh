cd /tmp && export TMPDIR=/tmp
run() { # label, dbg, shapes...
  label=$1; dbg=$2; shift 2
  rm -rf /tmp/sp; OMG_SCAN_DBG=$dbg timeout 100 rocprofv3 --kernel-trace -d /tmp/sp -o t --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/scan_probe.py sweep "$@" > /dev/null 2>&1
  echo "== $label (dbg $dbg): $@"; python3 $GRAFT_REPO_ROOT/tools/scan_trace.py /tmp/sp/t_kernel_trace.csv
}
timeout 200 python3 $GRAFT_REPO_ROOT/tools/scan_probe.py check 2>&1 | grep -c "e-1[56] $"
timeout 200 python3 $GRAFT_REPO_ROOT/tools/scan_probe.py check 2>&1 | grep -v "e-1[56] $"
for sh in 4x64x64 4x128x128 4x256x256 8x256x256 32x32x32 64x64x64 128x128x128 256x256x256; do run base 0 $sh; done

