#!/bin/bash
# builds the WORKING TREE's library with extra compile flags into tools/ab/libgr4pm_<name>.so (A/B timing)
# tools/build_variant.sh <name> [EXTRA="-D..."] [ABL="-D..."]     e.g. tools/build_variant.sh prio0 EXTRA=-DGR4PM_SERIAL_PRIO=0
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
mkdir -p $T/gr4-packet-modem_amd $T/include
cp -r $ROOT/gr4-packet-modem_amd/csrc $T/gr4-packet-modem_amd/csrc
cp $ROOT/include/*.h $T/include/
rm -f $T/gr4-packet-modem_amd/csrc/*.o
make -C $T/gr4-packet-modem_amd/csrc -j8 EXPERIMENTS=1 "$@" ../libgr4pm_hip.so > /dev/null 2>&1
mkdir -p $ROOT/tools/ab
cp $T/gr4-packet-modem_amd/libgr4pm_hip.so $ROOT/tools/ab/libgr4pm_$NAME.so
rm -rf $T
echo built tools/ab/libgr4pm_$NAME.so with "$@"
