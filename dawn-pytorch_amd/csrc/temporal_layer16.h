// Geometry of the window-tiled fused temporal layer (temporal_layer16.hip), shared with its launcher's schedule.
#pragma once
#define TL16_WAVES 8           /* 2 per SIMD: 256 registers each -- every weight fragment of a stage is requested a stage ahead */
#define TL16_ROWS 208          /* row slots of the X / K planes (a multiple of 16: the k-groups of a fragment read fall on disjoint banks) */
#define TL16_KEY_BLOCKS 6      /* 16-key blocks per 16-query tile: 16 + 2 win <= 96 */
#define TL16_LDS_BYTES 161856  /* X planes 79,872 + K planes 39,936 + V^T 39,936 + bias table copies 2,112 */
