#!/bin/bash
# A/B a list of build_variants/libvrt_<name>.so on the bench scenes: tools/ab_variants.sh name1 name2 ... [-- case ...]
names=(); cases=(config2 sunlit_1080 config4)
while [ $# -gt 0 ]; do if [ "$1" == "--" ]; then shift; cases=("$@"); break; fi; names+=("$1"); shift; done
for v in "${names[@]}"; do echo "== $v"; VRT_LIB_PATH=build_variants/libvrt_$v.so timeout -k 10 150 python tools/bench_scenes.py "${cases[@]}" 2>&1 | grep -o "name.*render_ms...[0-9.]*"; done
