/*
 * ref_luts.c -- caller-side EV tables for oracle/_ref/libmlvfs_ref.so.
 *
 * TEST INFRASTRUCTURE ONLY.  The reference's hot-path objects import
 * get_raw2ev / get_raw2evf / get_ev2raw from their caller (mlvfs/mlvfs.h:90-92;
 * defined in mlvfs/main.c:128-196, which needs <fuse.h> and therefore cannot be
 * built in this image).  These three accessors are the caller's part, supplied
 * here with the same formulas and the same libm (self-contained so that
 * the reference build links no oracle stage code).  Layout facts they have to honour: raw2ev is a 32768-entry table
 * pre-offset by (16384 - black) so that pixel values index it directly;
 * ev2raw is indexable from -10*32768 to 14*32768-1.
 */
#include <limits.h>
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

#define ORC_EV_RES    32768
#define ORC_MAX_BLACK 16384
#define ORC_EV2RAW_LO (-10 * ORC_EV_RES)

static int32_t tbl_raw2ev[2 * ORC_MAX_BLACK];
static double  tbl_raw2evf[2 * ORC_MAX_BLACK];
static int32_t tbl_ev2raw[24 * ORC_EV_RES];
static int ready;

static void init_tables(void)
{
    if (ready) return;
    /* linear value i lives at slot i + MAX_BLACK; slots below stay 0.  The
     * (int) cast of log2(0) = -inf is INT_MIN on x86-64 (cvttsd2si).           */
    for (int i = 0; i < 16384; i++) {
        double ev = log2((double)i) * ORC_EV_RES;
        tbl_raw2evf[i + ORC_MAX_BLACK] = ev;
        tbl_raw2ev[i + ORC_MAX_BLACK] = (i == 0) ? INT_MIN : (int32_t)ev;
    }
    for (int i = ORC_EV2RAW_LO; i < 14 * ORC_EV_RES; i++)
        tbl_ev2raw[i - ORC_EV2RAW_LO] = (int32_t)pow(2.0, (double)((float)i / ORC_EV_RES));
    ready = 1;
}

int *get_raw2ev(int black)
{
    init_tables();
    if (black > ORC_MAX_BLACK) { fprintf(stderr, "Black level too large for processing\n"); return NULL; }
    return (int *)tbl_raw2ev + (ORC_MAX_BLACK - black);
}

double *get_raw2evf(int black)
{
    init_tables();
    if (black > ORC_MAX_BLACK) { fprintf(stderr, "Black level too large for processing\n"); return NULL; }
    return tbl_raw2evf + (ORC_MAX_BLACK - black);
}

int *get_ev2raw(void)
{
    init_tables();
    return (int *)tbl_ev2raw - ORC_EV2RAW_LO;
}
