# Developer tool (GPU box): tools/mask_debug.py on the corner geometries of the matrix-pipe Tx-mask kernel (layout 15): shortest and
# longest symbols (P = 256 ... 336; the FIR tiling ends at cp + cs - tail_tx = 64), no fall tail, the longest fall tail, short and odd frames.
cd "$(dirname "$0")/.."
for g in "CP 256 4 16 0 0 0 21 2 11" "CP 256 2 16 0 0 4 21 2 12" "wtx 256 4 16 16 0 64 21 2 13" "WOLA 256 6 16 16 16 60 21 2 14" \
         "CPwtx 256 4 5 12 0 60 21 2 16" "wtx 256 4 2 16 0 32 21 2 18" "wtx 256 4 3 16 0 32 21 2 19"; do
  echo "== $g"
  MASK_DEBUG_TX=1 timeout -k 10 200 python tools/mask_debug.py $g 2>&1 | grep -v amdgpu.ids | grep "kernel\|stages\|^frame\|^[0-3] \[" | head -12
done
