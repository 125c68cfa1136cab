"""The algebra of the wrench-form constraint solver of the Stacking / Aligning engine (csrc/stack_step.h sk_solve_dual), in numpy.

The device solver never builds constraint rows: per contact it adds K = W' Hc W (6 x 6) and F = W' f (6) into the slot of its body pair, aggregates the slots
into eleven matrices / six net wrenches, and every dof lane builds its Hessian row as s_i' Keff(class i, class j) s_j.  This test restates that pipeline -
slots, signs, aggregates, the class table, the finger <-> finger pair that couples only the two slides - on random bodies and contacts and compares it with the
row form (J = sk_box_rows / sk_arm_rows written out, H = sum J' Hc J, g = - sum J' f) that the one-lane solver and the oracle use."""
import numpy as np


def test_wrench_form_equals_row_form():
    rng = np.random.default_rng(0)
    def rot():
        q = rng.normal(size=4); q /= np.linalg.norm(q); w,x,y,z = q
        return np.array([[1-2*(y*y+z*z),2*(x*y-z*w),2*(x*z+y*w)],[2*(x*y+z*w),1-2*(x*x+z*z),2*(y*z-x*w)],[2*(x*z-y*w),2*(y*z+x*w),1-2*(x*x+y*y)]])
    R = [rot() for _ in range(3)]; c = [rng.normal(size=3)*0.3 for _ in range(3)]
    z = [ (lambda v: v/np.linalg.norm(v))(rng.normal(size=3)) for _ in range(7)]; o = [rng.normal(size=3)*0.3 for _ in range(7)]
    ax = [ (lambda v: v/np.linalg.norm(v))(rng.normal(size=3)) for _ in range(2)]
    ref = np.array([0.1,0.05,-0.02])
    # J as the engine builds it
    def box_rows(b, p, fr):
        J = np.zeros((4,6)); r = p - c[b]
        for rr in range(3):
            f = fr[rr]; J[rr,:3] = f; J[rr,3:] = R[b].T @ np.cross(r, f)
        J[3,3:] = R[b].T @ fr[0]
        return J
    def arm_rows(f, p, fr):
        J = np.zeros((4,9))
        for k in range(7):
            col = np.cross(z[k], p - o[k])
            for rr in range(3): J[rr,k] = fr[rr] @ col
            J[3,k] = fr[0] @ z[k]
        if f >= 0:
            for rr in range(3): J[rr,7+f] = fr[rr] @ ax[f]
        return J
    def frame(n):
        n = n/np.linalg.norm(n); t = np.cross(n, [1,0,0.3]); t/=np.linalg.norm(t); return np.array([n,t,np.cross(n,t)])
    # bodies: 0..2 box, 3 static, 4/5 finger f, 6/7 tip f, 8 hand
    def fing(b): return -1 if b >= 8 else (b-4)&1
    cons = []
    pairs = [(3,0),(3,1),(3,2),(0,1),(0,2),(1,2),(0,4),(0,5),(1,6),(2,7),(1,8),(2,8),(4,5),(6,7),(2,4),(0,8)]
    for (a,b) in pairs:
        for rep in range(rng.integers(1,4)):
            p = rng.normal(size=3)*0.3; fr = frame(rng.normal(size=3))
            A = rng.normal(size=(4,4)); Hc = A@A.T; f = rng.normal(size=4)
            cons.append((a,b,p,fr,Hc,f))
    H = np.zeros((27,27)); g = np.zeros(27)
    x = rng.normal(size=27); Jx_ref = []
    for (a,b,p,fr,Hc,f) in cons:
        J = np.zeros((4,27))
        if b < 3: J[:,6*b:6*b+6] += box_rows(b,p,fr)
        else: J[:,18:] += arm_rows(fing(b),p,fr)
        if a < 3: J[:,6*a:6*a+6] -= box_rows(a,p,fr)
        elif a != 3: J[:,18:] -= arm_rows(fing(a),p,fr)
        H += J.T@Hc@J; g -= J.T@f; Jx_ref.append(J@x)
    # ---- wrench form
    S = np.zeros((27,6))
    for b in range(3):
        cp = c[b]-ref
        for k in range(3): S[6*b+k] = np.r_[0,0,0,np.eye(3)[k]]
        for k in range(3): w = R[b][:,k]; S[6*b+3+k] = np.r_[w, np.cross(cp,w)]
    for k in range(7): S[18+k] = np.r_[z[k], np.cross(o[k]-ref, z[k])]
    for f_ in range(2): S[25+f_] = np.r_[0,0,0,ax[f_]]
    def W(p,fr):
        pp = p-ref
        return np.array([np.r_[np.cross(pp,fr[0]),fr[0]], np.r_[np.cross(pp,fr[1]),fr[1]], np.r_[np.cross(pp,fr[2]),fr[2]], np.r_[fr[0],0,0,0]])
    # gen bodies: 0..2 box, 3 hand, 4 f1, 5 f2 ; slots
    def gb(b): return b if b<3 else (3 if b>=8 else 4+fing(b))
    def slot(a,b):
        A = -1 if a==3 else gb(a); B = gb(b)
        if A<0: return B                      # S_b 0..2
        if A<3 and B<3: return 3 + {(0,1):0,(0,2):1,(1,2):2}[(A,B)]   # BB
        if A<3 and B==3: return 6+A           # BH_b
        if A<3: return 9 + 2*A + (B-4)        # BF_{b,f}
        assert A==4 and B==5 or A==5 and B==4 or (A==B); return 15
    K = np.zeros((16,6,6)); F = np.zeros((16,6)); ffsign=[]
    # twists for J x check
    TW = np.zeros((4,6))
    for b in range(3): TW[b] = S[6*b:6*b+6].T@x[6*b:6*b+6]
    TW[3] = S[18:25].T@x[18:25]
    def twist(b):
        if b==3: return np.zeros(6)
        G = gb(b)
        if G<3: return TW[G]
        t = TW[3].copy()
        if G>=4: t[3:] += ax[G-4]*x[25+G-4]
        return t
    err=0
    for i,(a,b,p,fr,Hc,f) in enumerate(cons):
        w = W(p,fr); jx = w@(twist(b)-twist(a)); err=max(err,np.abs(jx-Jx_ref[i]).max())
        s = slot(a,b)
        sg = 1.0
        if s==15 and gb(a)==5: sg=-1.0   # finger-finger with A=f2,B=f1: store as (f1,f2) orientation: K same, F sign flips
        K[s] += w.T@Hc@w; F[s] += sg*(w.T@f)
    assert err < 1e-13
    BBi = {(0,1):3,(0,2):4,(1,2):5}
    def BB(a,b): return K[BBi[(min(a,b),max(a,b))]]
    Kd_b = [K[b] + sum(BB(b,o_) for o_ in range(3) if o_!=b) + K[6+b] + K[9+2*b] + K[10+2*b] for b in range(3)]
    K_barm = [K[6+b]+K[9+2*b]+K[10+2*b] for b in range(3)]
    Kdp_f = [sum(K[9+2*b+f_] for b in range(3)) for f_ in range(2)]
    K_rev = sum(K_barm); FF = K[15]; Kd_f = [Kdp_f[f_]+FF for f_ in range(2)]
    # net wrench on gen body: B side +, A side -
    N = np.zeros((6,6))
    for b in range(3):
        N[b] = F[b] + F[6+b]*(-1) + F[9+2*b]*(-1) + F[10+2*b]*(-1)
    N[0] += -F[3]-F[4]; N[1] += F[3]-F[5]; N[2] += F[4]+F[5]
    N[3] = F[6]+F[7]+F[8]
    N[4] = F[9]+F[11]+F[13] - F[15]; N[5] = F[10]+F[12]+F[14] + F[15]
    Ncls = [N[0],N[1],N[2],N[3]+N[4]+N[5],N[4],N[5]]
    def cls(i): return i//6 if i<18 else (3 if i<25 else 4+(i-25))
    H2 = np.zeros((27,27)); g2 = np.zeros(27)
    for i in range(27):
        ci = cls(i); g2[i] = -S[i]@Ncls[ci]
        for j in range(i+1):
            cj = cls(j)
            if ci<3:
                Ke = Kd_b[ci] if cj==ci else -BB(ci,cj)
            elif ci==3:
                Ke = -K_barm[cj] if cj<3 else K_rev
            else:
                f_ = ci-4
                if cj<3: Ke = -K[9+2*cj+f_]
                elif cj==3: Ke = Kdp_f[f_]
                elif cj==ci: Ke = Kd_f[f_]
                else: Ke = -FF
            H2[i,j] = S[i]@Ke@S[j]; H2[j,i]=H2[i,j]
    assert np.abs(H2 - H).max() < 1e-12 * np.abs(H).max() and np.abs(g2 - g).max() < 1e-12 * max(1.0, np.abs(g).max())

