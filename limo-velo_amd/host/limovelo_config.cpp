// limovelo_config.cpp — the default `Params Config` (reference src/main.cpp:14 defines the global itself) for host programs that do
// not: liblimovelo_shim_config.a, pulled by the linker only when nothing else on the link line defines the symbol.
#include "limovelo_shim.hpp"

struct Params Config;
