#!/bin/bash
# Build from a SNAPSHOT of the tree (so that sources may be edited while the ~7 minute merge-kernel compile runs: hipcc
# reads a header once for the device pass and again, minutes later, for the host pass -- an edit in between leaves an
# object whose host stubs name kernels its device code lacks, with a fresh timestamp that make then trusts), then copy
# the products back:   tools/safe_build.sh [make targets...]     (default: all clients)
set -e
root=/root/repo
snap=${CASK_SNAP:-/tmp/cask_snap}
mkdir -p $snap
# (no rsync in this image: cp -a --update keeps mtimes, so the snapshot's objects stay valid for unchanged sources)
for d in cask_amd include oracle tests tools Makefile __graft_entry__.py; do cp -a --update $root/$d $snap/; done
mkdir -p $snap/build && cp -a --update $root/build/obj $snap/build/ 2>/dev/null || true
cd $snap
targets=${*:-all clients}
make -j8 $targets
cp -a $snap/build/. $root/build/
cp -a $snap/cask_amd/lib/. $root/cask_amd/lib/
cp -a $snap/oracle/_build/. $root/oracle/_build/ 2>/dev/null || true
echo "safe_build: done"
