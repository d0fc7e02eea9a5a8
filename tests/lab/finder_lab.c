// tests/lab/finder_lab.c - CPU model of the region parse with different candidate structures (rows per tile, true last-R occurrences,
// round-granular hash chains), sized by the oracle's entropy stage.  Experiment bench kept beside the tests (it links the oracle); not collected by pytest, not part of the product.
// build: gcc -O2 -Ioracle -o /tmp/lab tests/lab/finder_lab.c oracle/libzso.so -Wl,-rpath,$PWD/oracle
// run:   /tmp/lab file H B R MODE HLOG LAZY frame [GRAN [LAZYT]]   (MODE 0 per-tile rows, 1 last R occurrences, 2 chains)
// finder lab: how much do window and candidate depth buy?  (CPU experiment, not part of the product)
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include "zso_enc.h"
typedef uint8_t u8_; 
static inline uint64_t rd64(const u8* p){uint64_t v;memcpy(&v,p,8);return v;}
static inline u32 hashN(const u8* p,int bytes,int log){ uint64_t v=rd64(p); v <<= (64-8*bytes); return (u32)((v*0x9E3779B185EBCA87ULL)>>(64-log)); }
static u32 mlen(const u8* a,const u8* b,const u8* end){ const u8* s=a; while(a<end && *a==*b){a++;b++;} return (u32)(a-s); }
// variant params
static int H, B, R, MODE, HLOG, LAZY, WIN, GRAN=256, LAZYT=1000;   // history bytes, block bytes, row depth, MODE: 0 = per-tile-first rows, 1 = true last-R occurrences; WIN: max distance for MODE 1/2
int main(int argc,char**argv){
  if(argc<9){fprintf(stderr,"usage: lab file H B R MODE HLOG LAZY frame\n");return 1;}
  FILE*f=fopen(argv[1],"rb"); fseek(f,0,SEEK_END); size_t n=ftell(f); fseek(f,0,SEEK_SET); u8* src=malloc(n+64); fread(src,1,n,f); fclose(f); memset(src+n,0,64);
  H=atoi(argv[2]); B=atoi(argv[3]); R=atoi(argv[4]); MODE=atoi(argv[5]); HLOG=atoi(argv[6]); LAZY=atoi(argv[7]); size_t frame=atol(argv[8]);
  if(argc>9) GRAN=atoi(argv[9]); if(argc>10) LAZYT=atoi(argv[10]);
  const int T=4096; u32* chain=malloc(4*(frame+64)); u32* tab2=malloc(4<<HLOG); size_t total=0, nseqTot=0; 
  u32 nb=1u<<HLOG; 
  u32* first=malloc(nb*4); u32* rows=malloc((size_t)nb*R*4); u32* head=malloc(nb*4);
  zso_seq* seqs=malloc(sizeof(zso_seq)*(B/3+16)); u8* lits=malloc(B+64); u8* out=malloc(B*2+1024);
  for(size_t fs=0; fs<n; fs+=frame){ size_t fe = fs+frame<n?fs+frame:n; total += 9; u32 rep[3]={1,4,8};
   for(size_t bs=fs; bs<fe; bs+=B){ size_t be=bs+B<fe?bs+B:fe; size_t lowl = bs-fs < (size_t)H ? fs : bs-H;   // window start
    // build structures from history
    memset(rows,0xFF,(size_t)nb*R*4); memset(head,0,nb*4);
    const u8* end=src+be;
    if(MODE==2){ memset(tab2,0xFF,4<<HLOG); for(size_t g0=lowl; g0<be; g0+=GRAN){ size_t g1=g0+GRAN<be?g0+GRAN:be; for(size_t p=g0;p<g1&&p+8<=be;p++) chain[p-fs]=tab2[hashN(src+p,5,HLOG)]; for(size_t p=g0;p<g1&&p+8<=be;p++) tab2[hashN(src+p,5,HLOG)]=(u32)(p-fs); } }
    size_t nseq=0, nl=0; size_t anchor=bs;
    // positions lowl..be in tiles aligned to bs (history tiles end at bs)
    // tile index relative: tile k covers [bs + (k)*T, ...) for k>=0 ; history tiles negative
    long kstart = -(long)((bs-lowl+T-1)/T);
    size_t p_next=bs; // parse cursor
    for(long k=kstart; ; k++){ long ts=(long)bs + k*T; if(ts>=(long)be) break; size_t t0 = ts<(long)lowl?lowl:(size_t)ts, t1 = (size_t)(ts+T)<be?(size_t)(ts+T):be;
      memset(first,0xFF,nb*4);
      // pass 1: first occurrence per bucket in the tile
      if(MODE==0) for(size_t p=t0;p<t1 && p+8<=be;p++){ u32 h=hashN(src+p,5,HLOG); if(first[h]==0xFFFFFFFFu) first[h]=(u32)(p-fs); }
      // parse positions of this tile (data tiles only)
      if(k>=0){ size_t p = p_next>t0?p_next:t0;
        while(p<t1 && p+8<=be){
          // best match at p
          u32 bestL=0, bestO=0;
          #define TRY(cp) do{ size_t c_=(cp); if(c_>=lowl && c_<pp_){ u32 l_=mlen(src+pp_,src+c_,end); if(l_>bl_ || (l_==bl_ && l_ && pp_-c_<bo_)){bl_=l_;bo_=(u32)(pp_-c_);} } }while(0)
          #define FIND(P,BL,BO) do{ size_t pp_=(P); u32 bl_=0,bo_=0; for(int d=1;d<=4;d++) if(pp_>=lowl+d && !memcmp(src+pp_,src+pp_-d,4)) { TRY(pp_-d); break;} \
             u32 h_=hashN(src+pp_,5,HLOG); if(MODE==0){ if(first[h_]!=0xFFFFFFFFu) TRY(fs+first[h_]); for(int r=0;r<R;r++) if(rows[(size_t)h_*R+r]!=0xFFFFFFFFu) TRY(fs+rows[(size_t)h_*R+r]); } \
             else if(MODE==1){ for(int r=0;r<R;r++){ u32 e=rows[(size_t)h_*R+r]; if(e!=0xFFFFFFFFu) TRY(fs+e);} } \
             else { u32 ch_=chain[pp_-fs]; for(int r=0;r<R && ch_!=0xFFFFFFFFu && fs+ch_>=lowl; r++){ TRY(fs+ch_); ch_=chain[ch_]; } } BL=bl_; BO=bo_; }while(0)
          if(MODE==1){ /* insert lazily: rows hold last R occurrences before p: maintained below */ }
          FIND(p,bestL,bestO);
          if(bestL<4 || (bestL==4 && bestO>=256)){ if(MODE==1){u32 h=hashN(src+p,5,HLOG); rows[(size_t)h*R+(head[h]++%R)]=(u32)(p-fs);} p++; continue; }
          if(LAZY){ for(;;){ if((int)bestL>=LAZYT) break; if(p+1+8>be) break; u32 l2,o2; if(MODE==1){u32 h=hashN(src+p,5,HLOG); rows[(size_t)h*R+(head[h]++%R)]=(u32)(p-fs);} FIND(p+1,l2,o2); if(l2<4){ if(MODE==1){ /* undo not needed */ } 
                 // no better
                 if(MODE==1){ /* p was inserted */ } goto nolazy; }
               int g1=(int)(bestL*4)-(int)(31-__builtin_clz(bestO+1))+4, g2=(int)(l2*4)-(int)(31-__builtin_clz(o2+1)); if(g2<=g1) goto nolazy; p++; bestL=l2; bestO=o2; }
             nolazy: ; }
          // backward extension
          { size_t c=p-bestO; while(p>anchor && c>lowl && src[p-1]==src[c-1]){p--;c--;bestL++;} }
          // emit
          u32 ll=(u32)(p-anchor); memcpy(lits+nl,src+anchor,ll); nl+=ll;
          // repcodes
          u32 ob; { u32 off=bestO; int ll0 = ll==0; if(!ll0 && off==rep[0]) ob=1; else if(off==rep[1]){ ob=ll0?1:2; } else if(off==rep[2]){ ob= ll0?2:3; } else if(ll0 && off==rep[0]-1 && off) ob=3; else ob=off+3;
             if(ob>3){rep[2]=rep[1];rep[1]=rep[0];rep[0]=off;} else { u32 rc = ob-1+ll0; if(rc){ u32 cur = rc==3?rep[0]-1:rep[rc]; if(rc>=2) rep[2]=rep[1]; rep[1]=rep[0]; rep[0]=cur; } } }
          seqs[nseq].offBase=ob; seqs[nseq].litLength=(u16)ll; seqs[nseq].mlBase=(u16)(bestL-3); nseq++;
          if(MODE==1){ for(size_t q=p+ (LAZY?1:0); q<p+bestL && q+8<=be; q++){ u32 h=hashN(src+q,5,HLOG); rows[(size_t)h*R+(head[h]++%R)]=(u32)(q-fs);} }
          p+=bestL; anchor=p;
        }
        p_next=p;
      } else if(MODE==1){ for(size_t p=t0;p<t1&&p+8<=be;p++){u32 h=hashN(src+p,5,HLOG); rows[(size_t)h*R+(head[h]++%R)]=(u32)(p-fs);} }
      // after the tile: its first occurrences go into the rows (slot = tile mod R)
      if(MODE==0){ long slot=((k%R)+R)%R; for(u32 h=0;h<nb;h++) if(first[h]!=0xFFFFFFFFu) rows[(size_t)h*R+slot]=first[h]; }
    }
    u32 ll=(u32)(be-anchor); memcpy(lits+nl,src+anchor,ll); nl+=ll;
    size_t sz = nseq||nl ? zso_entropy_block(out,B*2+1024,seqs,nseq,lits,nl,ZSO_greedy,be-bs) : 0;
    if(zso_isError(sz)) {fprintf(stderr,"entropy error\n"); return 2;}
    if(sz==0 || sz>=be-bs) sz=be-bs;
    total += 3+sz; nseqTot+=nseq;
   }
  }
  printf("H=%d B=%d R=%d MODE=%d HLOG=%d LAZY=%d frame=%zu : ratio %.4f  seqs %zu\n",H,B,R,MODE,HLOG,LAZY,frame,(double)total/n,nseqTot);
  return 0;
}
