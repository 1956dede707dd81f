// included by the node sources ("using namespace message_filters"), never used
#ifndef MOT_SHIM_MF_NS
#define MOT_SHIM_MF_NS
namespace message_filters {}
#endif
