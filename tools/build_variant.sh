#!/bin/bash
# A/B builds of the library without touching the tree: copies pyorc_amd/csrc + include/ to /tmp/variant_<name>, applies an optional
# patch (-p file, relative to the repo root, `patch -p0` form with paths pyorc_amd/csrc/...) and extra -D flags, builds only the
# translation units named by UNITS (default: piv_fft32 piv_fft64 -- the rest are taken from the tree's objects), links
# build/ab/lib_<name>.so.  Load it with LSPIV_LIBRARY=build/ab/lib_<name>.so (exempt from the stale check).
#   usage: tools/build_variant.sh <name> [-r git-rev] [-p patchfile] [-DFOO=1 ...]      env: UNITS="piv_fft64 lspiv_api"
# -r: the kernel headers (common.h fft_regs.h piv_fft_impl.h) of that revision instead of the tree's.  Only the UNITS are recompiled
# (the other objects are the tree's, touched so that make leaves them alone).
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
patchf=""; rev=""
if [ "$1" = "-r" ]; then rev=$2; shift 2; fi
if [ "$1" = "-p" ]; then patchf=$2; shift 2; fi
V=/tmp/variant_$name
rm -rf $V; mkdir -p $V/pyorc_amd $V/include $R/build/ab
cp -r $R/pyorc_amd/csrc $V/pyorc_amd/csrc
cp $R/include/lspiv.h $V/include/
if [ -n "$rev" ]; then for h in common.h fft_regs.h piv_fft_impl.h; do git -C $R show $rev:pyorc_amd/csrc/$h > $V/pyorc_amd/csrc/$h; done; fi
if [ -n "$patchf" ]; then (cd $V && patch -p0 --no-backup-if-mismatch < "$R/$patchf"); fi
cd $V/pyorc_amd/csrc
touch *.o
for u in ${UNITS:-piv_fft32 piv_fft64}; do rm -f $u.o; done
make -j8 HIPFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -Wall -Wno-unused-function -Wno-unused-value $*" TARGET=$R/build/ab/lib_$name.so >/dev/null
echo "built build/ab/lib_$name.so"
