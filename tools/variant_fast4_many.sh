#!/bin/bash
# Build several diagnostic variants of the pair kernel in parallel: tools/variant_fast4_many.sh name1:"flags" name2:"flags" ...
cd "$(dirname "$0")/.."
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  ( tools/variant_fast4.sh $name $flags > /tmp/variant_$name.log 2>&1 || echo "variant $name FAILED" ) &
done
wait
ls -la walnuts_amd/libwalnuts_amd_*.so
