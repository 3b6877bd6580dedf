for L in 1 2; do for H in 8 16 32; do for B in 8 16 32; do
V=$(( ((L+1)<<8) | ((H+1)<<16) | (B<<24) ))
R=$(timeout -k 10 120 python tools/share_cost.py 8 $V 2>/dev/null | grep "rank 0" | sed 's/.*share in one launch \([0-9.]*\) ms.*/\1/')
F=$(timeout -k 10 120 python tools/share_cost.py 8 $V 2>/dev/null | grep "rank 0" | sed 's/.*full frame \([0-9.]*\) ms.*/\1/')
echo "leave $L heavyMin $H leafBias $B : share $R ms, full frame $F ms"
done; done; done
