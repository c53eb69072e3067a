// test scaffold: forwards to the declarations in ../../reference_layer_decls.hpp (see there)
#include "../../reference_layer_decls.hpp"
