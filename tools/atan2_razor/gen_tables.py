from decimal import Decimal, getcontext
import struct
getcontext().prec = 60
PI = Decimal("3.14159265358979323846264338327950288419716939937510582097494")
def dsin(x):
    # Taylor, x small-ish after reduction to [-pi/4, pi/4] by caller
    t=x; s=x; n=1
    while abs(t) > Decimal(10)**-55:
        t = -t*x*x/((2*n)*(2*n+1)); s+=t; n+=1
    return s
def dcos(x):
    t=Decimal(1); s=Decimal(1); n=1
    while abs(t) > Decimal(10)**-55:
        t = -t*x*x/((2*n-1)*(2*n)); s+=t; n+=1
    return s
def sincos_deg(k):
    # exact symmetries to keep arguments small
    k%=360
    q,r=divmod(k,90)
    x=Decimal(r)*PI/180
    if r<=45: s,c=dsin(x),dcos(x)
    else:
        y=Decimal(90-r)*PI/180; s,c=dcos(y),dsin(y)
    for _ in range(q): s,c=c,-s
    return s,c
def split(v):
    hi=float(v); lo=float(v-Decimal(hi)); return hi,lo
rows=[]
for k in range(181):
    s,c=sincos_deg(k); phi=Decimal(k)*PI/180
    rows.append(split(phi)+split(c)+split(s))
import os,sys
OUT=sys.argv[1] if len(sys.argv)>1 else "tab.h"
with open(OUT,"w") as f:
    f.write("/* k degrees, k = 0..180: (phi_hi, phi_lo, cos_hi, cos_lo, sin_hi, sin_lo), double-double, generated with 60-digit decimals */\n")
    f.write("static const double kDegDD[181][6] = {\n")
    for r in rows: f.write("  {"+", ".join(float.hex(x) for x in r)+"},\n")
    f.write("};\n")
print(rows[105], rows[90])
