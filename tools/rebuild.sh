#!/bin/bash
# rebuilds the library and the CLI in-tree (errors shown)
set -e
cd "$(dirname "$0")/.."
python -c "from soapnuke_amd import build; build.build(force=True); build.build_host(force=True)"
ls -la soapnuke_amd/libsnk_filter.so soapnuke_amd/SOAPnuke
