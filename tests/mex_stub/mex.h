/* mex.h -- STUB, test infrastructure only.  Declares the handful of MATLAB MEX API entry points that
 * matlab/nrldpc_mex.cpp uses, with the signatures of MathWorks' documented C Matrix / MEX API, so that the CPU test suite
 * can compile-check the gateway (g++ -fsyntax-only -Wall -Wextra) in an image that has no MATLAB.  Nothing here is ever
 * linked or run; a real build uses MATLAB's own mex.h (`mex -I<repo>/include ... nrldpc_mex.cpp`). */
#ifndef NRLDPC_TEST_MEX_STUB_H
#define NRLDPC_TEST_MEX_STUB_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef struct mxArray_tag mxArray;
typedef size_t mwSize;
typedef enum { mxUNKNOWN_CLASS = 0, mxDOUBLE_CLASS = 6, mxUINT8_CLASS = 9, mxINT32_CLASS = 12 } mxClassID;
typedef enum { mxREAL = 0, mxCOMPLEX = 1 } mxComplexity;
bool mxIsChar(const mxArray*);
bool mxIsDouble(const mxArray*);
bool mxIsSingle(const mxArray*);
bool mxIsComplex(const mxArray*);
int mxGetString(const mxArray*, char*, mwSize);
double mxGetScalar(const mxArray*);
size_t mxGetM(const mxArray*);
size_t mxGetN(const mxArray*);
size_t mxGetNumberOfElements(const mxArray*);
double* mxGetPr(const mxArray*);
void* mxGetData(const mxArray*);
mxArray* mxCreateDoubleScalar(double);
mxArray* mxCreateDoubleMatrix(mwSize, mwSize, mxComplexity);
mxArray* mxCreateNumericMatrix(mwSize, mwSize, mxClassID, mxComplexity);
void mxDestroyArray(mxArray*);
void mexErrMsgIdAndTxt(const char*, const char*, ...) __attribute__((noreturn));
void mexLock(void);
void mexUnlock(void);
int mexAtExit(void (*)(void));
void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]);
#ifdef __cplusplus
}
#endif
#endif
