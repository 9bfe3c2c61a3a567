// libpcr.so -- host entry point of the 3x3 solve (the math lives in solve_core.h, shared with the device loop).
#include <string.h>
#include "solve_core.h"

extern "C" int pcr_solve_rigid(const double partials[17], const double pivot_p[3], const double pivot_q[3],
                               double R[9], double T[3], double Rc[9], double Tc[3], double *mse) {
    PCR_REQUIRE(partials && pivot_p && pivot_q && R && T, "pcr_solve_rigid: NULL argument");
    const double n = partials[0];
    if (!(n >= 1.0)) {
        pcr::set_last_error("pcr_solve_rigid: no correspondence survived (n=%g)", n);
        return PCR_E_EMPTY;
    }
    pcr::solve_rigid_core(partials, pivot_p, pivot_q, R, T, Rc, Tc, mse);
    return PCR_OK;
}
