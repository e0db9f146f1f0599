/* TEST INFRASTRUCTURE ONLY (oracle).  Jacobian short-Weierstrass group law, y^2 = x^3 + b,
 * instantiated twice by oracle.c (G1 over Fp, G2 over Fp2).  Restates the semantics of
 * kyber.Point Add/Mul/Neg/Null as the reference uses them (curve.go:25-45, algebra.go:356,
 * groth16.go:138,149-152,189-200); the arithmetic itself lives upstream (kilic/bls12-381).
 *
 * Required macros: CN(name) F FADD FSUB FMUL FSQR FNEG FINV FISZERO FEQ FONE FZERO FB
 */

typedef struct { F x, y; int inf; } CN(aff);
typedef struct { F x, y, z; } CN(jac); /* z == 0 <=> identity */

static void CN(jac_set_inf)(CN(jac)* p) { p->x = FONE; p->y = FONE; p->z = FZERO; }
static int CN(jac_is_inf)(const CN(jac)* p) { return FISZERO(&p->z); }

static void CN(from_aff)(CN(jac)* o, const CN(aff)* a) {
    if (a->inf) { CN(jac_set_inf)(o); return; }
    o->x = a->x; o->y = a->y; o->z = FONE;
}

static void CN(to_aff)(CN(aff)* o, const CN(jac)* p) {
    if (CN(jac_is_inf)(p)) { o->inf = 1; o->x = FZERO; o->y = FZERO; return; }
    F zi, zi2, zi3;
    FINV(&zi, &p->z);
    FSQR(&zi2, &zi);
    FMUL(&zi3, &zi2, &zi);
    FMUL(&o->x, &p->x, &zi2);
    FMUL(&o->y, &p->y, &zi3);
    o->inf = 0;
}

/* dbl-2009-l (a = 0) */
static void CN(dbl)(CN(jac)* o, const CN(jac)* p) {
    if (CN(jac_is_inf)(p)) { *o = *p; return; }
    F A, B, C, D, E, Fv, t;
    FSQR(&A, &p->x);
    FSQR(&B, &p->y);
    FSQR(&C, &B);
    FADD(&t, &p->x, &B); FSQR(&t, &t); FSUB(&t, &t, &A); FSUB(&t, &t, &C); FADD(&D, &t, &t);
    FADD(&E, &A, &A); FADD(&E, &E, &A);
    FSQR(&Fv, &E);
    F z3; FMUL(&z3, &p->y, &p->z); FADD(&z3, &z3, &z3);
    FSUB(&o->x, &Fv, &D); FSUB(&o->x, &o->x, &D);
    FSUB(&t, &D, &o->x); FMUL(&t, &E, &t);
    FADD(&C, &C, &C); FADD(&C, &C, &C); FADD(&C, &C, &C);
    FSUB(&o->y, &t, &C);
    o->z = z3;
}

/* add-2007-bl with the exceptional cases handled */
static void CN(add)(CN(jac)* o, const CN(jac)* p, const CN(jac)* q) {
    if (CN(jac_is_inf)(p)) { *o = *q; return; }
    if (CN(jac_is_inf)(q)) { *o = *p; return; }
    F z1z1, z2z2, u1, u2, s1, s2, h, i, j, r, v, t;
    FSQR(&z1z1, &p->z); FSQR(&z2z2, &q->z);
    FMUL(&u1, &p->x, &z2z2); FMUL(&u2, &q->x, &z1z1);
    FMUL(&s1, &p->y, &q->z); FMUL(&s1, &s1, &z2z2);
    FMUL(&s2, &q->y, &p->z); FMUL(&s2, &s2, &z1z1);
    FSUB(&h, &u2, &u1);
    FSUB(&r, &s2, &s1);
    if (FISZERO(&h)) {
        if (FISZERO(&r)) { CN(dbl)(o, p); return; }
        CN(jac_set_inf)(o); return;
    }
    FADD(&r, &r, &r);
    FADD(&i, &h, &h); FSQR(&i, &i);
    FMUL(&j, &h, &i);
    FMUL(&v, &u1, &i);
    F x3, y3, z3;
    FSQR(&x3, &r); FSUB(&x3, &x3, &j); FSUB(&x3, &x3, &v); FSUB(&x3, &x3, &v);
    FSUB(&t, &v, &x3); FMUL(&y3, &r, &t);
    FMUL(&t, &s1, &j); FADD(&t, &t, &t); FSUB(&y3, &y3, &t);
    FADD(&z3, &p->z, &q->z); FSQR(&z3, &z3); FSUB(&z3, &z3, &z1z1); FSUB(&z3, &z3, &z2z2);
    FMUL(&z3, &z3, &h);
    o->x = x3; o->y = y3; o->z = z3;
}

/* mixed add: q affine */
static void CN(madd)(CN(jac)* o, const CN(jac)* p, const CN(aff)* q) {
    if (q->inf) { *o = *p; return; }
    if (CN(jac_is_inf)(p)) { CN(from_aff)(o, q); return; }
    CN(jac) qq; CN(from_aff)(&qq, q);
    CN(add)(o, p, &qq);
}

static void CN(neg_aff)(CN(aff)* o, const CN(aff)* a) {
    *o = *a;
    if (!a->inf) FNEG(&o->y, &a->y);
}

/* Point.Mul(s, p): MSB-first double-and-add over the canonical scalar (4x64 plain limbs) */
static void CN(mul)(CN(jac)* o, const uint64_t k[4], const CN(aff)* p) {
    CN(jac) acc; CN(jac_set_inf)(&acc);
    int started = 0;
    for (int bit = 255; bit >= 0; bit--) {
        if (started) CN(dbl)(&acc, &acc);
        if ((k[bit >> 6] >> (bit & 63)) & 1) { CN(madd)(&acc, &acc, p); started = 1; }
    }
    *o = acc;
}

static int CN(on_curve)(const CN(aff)* a) {
    if (a->inf) return 1;
    F l, r;
    FSQR(&l, &a->y);
    FSQR(&r, &a->x); FMUL(&r, &r, &a->x); { F b = FB; FADD(&r, &r, &b); }
    return FEQ(&l, &r);
}

/* batch normalisation (Montgomery's trick) */
static void CN(batch_to_aff)(CN(aff)* out, const CN(jac)* in, size_t n) {
    F* pref = (F*)malloc(sizeof(F) * (n ? n : 1));
    F acc = FONE;
    for (size_t i = 0; i < n; i++) {
        pref[i] = acc;
        if (!CN(jac_is_inf)(&in[i])) FMUL(&acc, &acc, &in[i].z);
    }
    F inv; FINV(&inv, &acc);
    for (size_t i = n; i-- > 0;) {
        if (CN(jac_is_inf)(&in[i])) { out[i].inf = 1; out[i].x = FZERO; out[i].y = FZERO; continue; }
        F zi, zi2, zi3;
        FMUL(&zi, &inv, &pref[i]);
        FMUL(&inv, &inv, &in[i].z);
        FSQR(&zi2, &zi); FMUL(&zi3, &zi2, &zi);
        FMUL(&out[i].x, &in[i].x, &zi2);
        FMUL(&out[i].y, &in[i].y, &zi3);
        out[i].inf = 0;
    }
    free(pref);
}

/* ---- Pippenger bucket MSM (CPU baseline B1; unsigned c-bit windows) ---- */
typedef struct {
    const uint64_t* k; /* n x 4 plain limbs */
    const CN(aff)* pts;
    size_t n;
    int c, nwin, tid, nthreads;
    CN(jac)* win_sums;
} CN(pip_job);

static void* CN(pip_worker)(void* arg) {
    CN(pip_job)* job = (CN(pip_job)*)arg;
    const int c = job->c;
    const size_t nb = ((size_t)1 << c) - 1;
    CN(jac)* buckets = (CN(jac)*)malloc(sizeof(CN(jac)) * nb);
    for (int w = job->tid; w < job->nwin; w += job->nthreads) {
        for (size_t b = 0; b < nb; b++) CN(jac_set_inf)(&buckets[b]);
        const int lo = w * c;
        for (size_t i = 0; i < job->n; i++) {
            const uint64_t* k = job->k + 4 * i;
            uint64_t d = k[lo >> 6] >> (lo & 63);
            if ((lo & 63) + c > 64 && (lo >> 6) + 1 < 4) d |= k[(lo >> 6) + 1] << (64 - (lo & 63));
            d &= nb;
            if (d) CN(madd)(&buckets[d - 1], &buckets[d - 1], &job->pts[i]);
        }
        CN(jac) run, acc; CN(jac_set_inf)(&run); CN(jac_set_inf)(&acc);
        for (size_t b = nb; b-- > 0;) {
            CN(add)(&run, &run, &buckets[b]);
            CN(add)(&acc, &acc, &run);
        }
        job->win_sums[w] = acc;
    }
    free(buckets);
    return NULL;
}

static void CN(pippenger)(CN(jac)* out, const uint64_t* k, const CN(aff)* pts, size_t n, int threads) {
    int c = 4;
    if (n >= 32) { c = 0; size_t t = n; while (t > 1) { t >>= 1; c++; } c = c > 6 ? c - 3 : 3; if (c > 16) c = 16; }
    int nwin = (255 + c) / c;
    if (threads < 1) threads = 1;
    if (threads > nwin) threads = nwin;
    CN(jac)* ws = (CN(jac)*)malloc(sizeof(CN(jac)) * nwin);
    CN(pip_job)* jobs = (CN(pip_job)*)malloc(sizeof(CN(pip_job)) * threads);
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * threads);
    for (int t = 0; t < threads; t++) {
        jobs[t] = (CN(pip_job)){k, pts, n, c, nwin, t, threads, ws};
        if (threads == 1) CN(pip_worker)(&jobs[t]);
        else pthread_create(&th[t], NULL, CN(pip_worker), &jobs[t]);
    }
    if (threads > 1) for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
    CN(jac) acc; CN(jac_set_inf)(&acc);
    for (int w = nwin - 1; w >= 0; w--) {
        for (int i = 0; i < c; i++) CN(dbl)(&acc, &acc);
        CN(add)(&acc, &acc, &ws[w]);
    }
    *out = acc;
    free(ws); free(jobs); free(th);
}
