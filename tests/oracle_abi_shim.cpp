// the one entry of the ABI the oracle does not export (see oracle_abi_shim.h)
struct obvi_ba_handle;
extern "C" const char* oracle_ba_last_error(const obvi_ba_handle*) { return "(oracle: no error text)"; }
