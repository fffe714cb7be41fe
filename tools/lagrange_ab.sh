set -e
B=mpc-jellyfish_amd/mzk_prove
a=$($B 0 turbo 4096 1 | python3 -c "import json,sys; print(json.load(sys.stdin)['proof_hex'])")
b=$($B 0 turbo 4096 1 --lagrange | python3 -c "import json,sys; print(json.load(sys.stdin)['proof_hex'])")
c=$(MZK_VIRTUAL_DEVICES=4 $B 0 turbo 4096 1 --lagrange --gpus 4 --check-agree | python3 -c "import json,sys; print(json.load(sys.stdin)['proof_hex'])")
d=$($B 1 ultra 4096 1 | python3 -c "import json,sys; print(json.load(sys.stdin)['proof_hex'])")
e=$($B 1 ultra 4096 1 --lagrange | python3 -c "import json,sys; print(json.load(sys.stdin)['proof_hex'])")
[ "$a" = "$b" ] && [ "$a" = "$c" ] && [ "$d" = "$e" ] && echo "same proof bytes: coefficient commit / lagrange / lagrange on 4 virtual devices; ultra too"
for f in "" "--lagrange"; do $B 0 turbo 1048576 8 $f | cut -c1-560; done
$B 1 ultra 1048576 8 --lagrange | cut -c1-700
