/* placeholder until the mini-Box2D restatement lands */
#include "oracle.h"
#include <stdlib.h>
struct ora_lunarlander { int dummy; };
ora_lunarlander *ora_lunarlander_new(float g, int w, float wp, float tp, int *st) { (void)g;(void)w;(void)wp;(void)tp; if (st) *st = ORA_BAD_CONFIG; return NULL; }
void ora_lunarlander_free(ora_lunarlander *e) { free(e); }
int ora_lunarlander_reset(ora_lunarlander *e, const float *a, const float *b, int32_t c, int32_t d, const float *f, float *o) { (void)e;(void)a;(void)b;(void)c;(void)d;(void)f;(void)o; return ORA_BAD_CONFIG; }
int ora_lunarlander_reset_deterministic(ora_lunarlander *e, float *o) { (void)e;(void)o; return ORA_BAD_CONFIG; }
int ora_lunarlander_step(ora_lunarlander *e, uint32_t a, const float *d, float *o, ora_stepinfo *s) { (void)e;(void)a;(void)d;(void)o;(void)s; return ORA_BAD_CONFIG; }
void ora_lunarlander_set_state(ora_lunarlander *e, const float *r, int a, int b) { (void)e;(void)r;(void)a;(void)b; }
int ora_lunarlander_state_floats(void) { return 0; }
void ora_lunarlander_export(const ora_lunarlander *e, float *b) { (void)e;(void)b; }
void ora_lunarlander_import(ora_lunarlander *e, const float *b) { (void)e;(void)b; }
