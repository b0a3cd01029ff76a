#!/bin/bash
# builds the library of a git revision into tools/ab/libgr4pm_<name>.so for A/B timing (tools/ab_correlate.sh)
# tools/build_ab.sh <rev> <name>
set -e
REV=${1:-HEAD}; NAME=${2:-prev}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
mkdir -p $T/pkg/csrc $T/include
git -C $ROOT archive $REV gr4-packet-modem_amd/csrc include | tar -x -C $T
mkdir -p $T/gr4-packet-modem_amd/csrc
make -C $T/gr4-packet-modem_amd/csrc -j6 > /dev/null
mkdir -p $ROOT/tools/ab
cp $T/gr4-packet-modem_amd/libgr4pm_hip.so $ROOT/tools/ab/libgr4pm_$NAME.so
rm -rf $T
echo built tools/ab/libgr4pm_$NAME.so from $REV
