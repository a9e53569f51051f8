// explicit instantiations of the key-type launchers for u32 keys
#include "keyed_impl.h"
ILLICO_KEYED_INSTANCES(, u32)
