#ifndef LVREF_STD_FLOAT32_STUB
#define LVREF_STD_FLOAT32_STUB
namespace std_msgs { struct Float32 { float data = 0; }; }
#endif
