import pandas as pd, sys
k=pd.read_csv(sys.argv[1])
k['dur']=(k.End_Timestamp-k.Start_Timestamp)/1e3
k['nm']=k.Kernel_Name.str.replace('void ','').str.replace('(anonymous namespace)::','',regex=False).str.split('(').str[0].str[:40]
g=k.groupby('nm').dur.agg(['count','sum','mean']).sort_values('sum',ascending=False)
print(g.head(22).to_string())
