# per-kernel A/B of two library builds on one box (ab/old.so, ab/new.so): scripts/dw_form_ab.py under each
L=mammo_clip_amd/lib/libmammoclip_hip.so
cp $L /tmp/keep.so
for v in old new old new; do cp ab/$v.so $L; echo "== $v"; python scripts/dw_form_ab.py 2>/dev/null | grep "k5 s1\|k5 s2"; done
cp /tmp/keep.so $L
