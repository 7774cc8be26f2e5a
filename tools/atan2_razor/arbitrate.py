from decimal import Decimal, getcontext
getcontext().prec=70
import math
exec(open(__import__('os').path.join(__import__('os').path.dirname(__import__('os').path.abspath(__file__)), 'gen_tables.py')).read().split("rows=[]")[0])
cases=[(8,"-0x1.dce6f693f96d7p+3","0x1.a82abc3494767p+6","-0x1.1df46a2529d38p-3","-0x1.1df46a2529d39p-3"),
(49,"0x1.383559c6c888ap+6","0x1.0f660c206abeep+6","0x1.b5de4288e80cp-1","0x1.b5de4288e80bfp-1"),
(17,"-0x1.a0181acb41b5ap+6","0x1.543ee40c79b39p+8","-0x1.2fd3b0c77be9ap-2","-0x1.2fd3b0c77be99p-2"),
(25,"0x1.6d2a0f34aa0ddp+5","0x1.878c762f62102p+6","0x1.becde5da11f48p-2","0x1.becde5da11f49p-2")]
for k,ys,xs,gs,rs in cases:
    y=Decimal(abs(float.fromhex(ys))); x=Decimal(float.fromhex(xs))
    s,c=sincos_deg(k); phi=Decimal(k)*PI/180
    N=y*c-x*s; D=x*c+y*s; t=N/D
    delta=t-t**3/3
    theta=phi+delta
    g=abs(float.fromhex(gs)); r=abs(float.fromhex(rs))
    # which double is nearest
    dg=abs(Decimal(g)-theta); dr=abs(Decimal(r)-theta)
    ulp=Decimal(math.ulp(g))
    print(k,"err glibc %.3f ulp, razor %.3f ulp"%(float(dg/ulp),float(dr/ulp)), "-> correct:", "glibc" if dg<dr else "razor")
