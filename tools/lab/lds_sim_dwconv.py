"""LDS bank-conflict search for the channel-plane layout of csrc/dwconv_mfma.h (development aid): plane / row pitches and the item order of the
transposition that keep the ds_read_b64 of the MFMA phase conflict free and the ds_write_b32 of the transposition at 2-way."""
import itertools
def conflicts_b32(addrs):  # list of 64 dword addresses; ds_*_b32: 2 groups of 32 lanes, bank = addr % 32; identical addresses broadcast (reads)
    tot=0
    for g in range(2):
        a=addrs[32*g:32*g+32]
        banks={}
        for x in a:
            if x is None: continue
            banks.setdefault(x%32,set()).add(x)
        tot+=max([len(v) for v in banks.values()] or [1])
    return tot  # cycles (2 = conflict free)
def sim(R, PWd, OP, CHS):
    # read pattern: lanes l: b=l>>2, i=l&3 ; two dwords (d, d+1)
    worst_r=0
    for kh in range(1):
        for m in range(1):
            for dd in (0,1):
                addrs=[ (l>>2)*CHS + ((l&3)+kh)*PWd + 2*m + dd for l in range(64)]
                worst_r=max(worst_r, conflicts_b32(addrs))
    # write pattern: item t = wave*64 + lane: o = t % OP, pp = (t//OP) % PP, r = t // (OP*PP)
    PP=PWd
    tot_w=0; n=0
    for wave in range(4):
        for e in range(8):
            addrs=[]
            for l in range(64):
                t=wave*64+l
                o=t%OP; pp=(t//OP)%PP; r=t//(OP*PP)
                addrs.append((8*o+e)*CHS + r*PWd + pp)
            tot_w+=conflicts_b32(addrs); n+=1
    return worst_r, tot_w/n
for (name,R,PWd,OP) in [("C96 TH8 TW16",14,12,12),("C192/384 pass96.. TH8",14,12,12),("C384 pass192 TH4",10,12,24)]:
    best=[]
    for pad in range(0,40):
        CHS=R*PWd+pad
        r,w=sim(R,PWd,OP,CHS)
        best.append((r+w, r, w, pad))
    best.sort()
    print(name, [(round(a,1),r,round(w,1),p) for a,r,w,p in best[:6]])
print("---- extended search")
def sim2(R, PWd0, OP, pad, rowpad, OL):
    PWd=PWd0+rowpad; CHS=R*PWd+pad
    worst_r=0
    for dd in (0,1):
        addrs=[ (l>>2)*CHS + ((l&3))*PWd + dd for l in range(64)]
        worst_r=max(worst_r, conflicts_b32(addrs))
    PP=PWd0
    tot_w=0; n=0
    # item t: ol = t % OL ; pp = (t//OL) % PP ; r = (t//(OL*PP)) % R ; oh = t // (OL*PP*R) ; octet = oh*OL + ol
    for wave in range(6):
        for e in range(8):
            addrs=[]
            for l in range(64):
                t=wave*64+l
                ol=t%OL; pp=(t//OL)%PP; r=(t//(OL*PP))%R; oh=t//(OL*PP*R)
                o=oh*OL+ol
                addrs.append((8*o+e)*CHS + r*PWd + pp)
            tot_w+=conflicts_b32(addrs); n+=1
    return worst_r, tot_w/n
for (name,R,PWd0,OP) in [("C96 TH8 TW16",14,12,12),("C384 pass192 TH4",10,12,24)]:
    res=[]
    for OL in (12,6,4,3,2,1):
        if OP%OL: continue
        for rowpad in range(0,5):
            for pad in range(0,33):
                r,w=sim2(R,PWd0,OP,pad,rowpad,OL)
                res.append((r*2+w, r, round(w,2), OL, rowpad, pad, (R*(PWd0+rowpad)+pad)*4))
    res.sort()
    print(name); [print("   score %.1f read %d write %.2f OL %d rowpad %d pad %d bytes/plane %d"%x) for x in res[:8]]
print("---- per OL best")
for (name,R,PWd0,OP) in [("C96 TH8 TW16",14,12,12),("C384 pass192 TH4",10,12,24)]:
    for OL in (12,6,4,3,2):
        if OP%OL: continue
        res=[]
        for rowpad in range(0,5):
            for pad in range(0,33):
                r,w=sim2(R,PWd0,OP,pad,rowpad,OL)
                res.append((r*2+w, r, round(w,2), OL, rowpad, pad, (R*(PWd0+rowpad)+pad)*4))
        res.sort()
        print(name, "OL",OL, "best: score %.1f read %d write %.2f OL %d rowpad %d pad %d bytes/plane %d"%res[0])
print("---- ds_read_b64 model (64 banks, 2 groups of 32 lanes, 2 dwords per lane, 8-byte aligned)")
def conf_b64(addrs):
    tot=0
    for g in range(2):
        a=addrs[32*g:32*g+32]
        banks={}
        for x in a:
            for d in (0,1):
                banks.setdefault((x+d)%64,set()).add(x+d)
        tot+=max(len(v) for v in banks.values())
    return tot
def conf_w32(addrs): return conflicts_b32(addrs)
for (name,R,PWd0,OP) in [("C96 TH8 TW16",14,12,12),("C384 pass192 TH4",10,12,24)]:
    for OL in (12,6,4):
        res=[]
        for rowpad in range(0,9,2):
            for pad in range(0,65,2):
                PWd=PWd0+rowpad; CHS=R*PWd+pad
                rd=conf_b64([ (l>>2)*CHS + (l&3)*PWd for l in range(64)])
                PP=PWd0; tot=0;n=0
                for wave in range(6):
                    for e in range(8):
                        addrs=[]
                        for l in range(64):
                            t=wave*64+l
                            ol=t%OL; pp=(t//OL)%PP; r=(t//(OL*PP))%R; oh=t//(OL*PP*R)
                            addrs.append((8*(oh*OL+ol)+e)*CHS + r*PWd + pp)
                        tot+=conf_w32(addrs); n+=1
                res.append((rd*1.0+tot/n*0.5, rd, round(tot/n,2), rowpad, pad, CHS*4))
        res.sort()
        print(name,"OL",OL,"best (score, read cyc [2=free], write cyc [2=free], rowpad, pad, bytes/plane):", res[:3])
print("---- order: ol(4) fastest, pp, oh, r   (row-major over octet groups)")
for (name,R,PWd0,OP) in [("C96 TH8 TW16",14,12,12),("C192pass TH4",10,12,24)]:
    for OL in (4,6,12):
        OH=OP//OL
        res=[]
        for rowpad in range(0,9,2):
            for pad in range(0,65,2):
                PWd=PWd0+rowpad; CHS=R*PWd+pad
                rd=conf_b64([ (l>>2)*CHS + (l&3)*PWd for l in range(64)])
                PP=PWd0; tot=0;n=0
                for wave in range(9):
                    for e in range(8):
                        addrs=[]
                        for l in range(64):
                            t=wave*64+l
                            ol=t%OL; pp=(t//OL)%PP; oh=(t//(OL*PP))%OH; r=t//(OL*PP*OH)
                            addrs.append((8*(oh*OL+ol)+e)*CHS + r*PWd + pp)
                        tot+=conf_w32(addrs); n+=1
                res.append((rd*1.0+tot/n*0.5, rd, round(tot/n,2), rowpad, pad, CHS*4))
        res.sort()
        print(name,"OL",OL,"best:", res[:2])
